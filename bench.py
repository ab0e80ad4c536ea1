"""bench.py -- Stage-I training rays/sec on MI355X (BASELINE.json metric).  One JSON line on rank 0.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one full training step of the GlossySynthetic 'bell' Stage-I shape configuration (configs[1]): 4096 rays per GPU x
(64 coarse + 64 importance + 32 background) samples, hierarchical sampling (112 no-grad SDF evals/ray), render forward, loss,
backward including the second-order SDF term, flat RCCL gradient all-reduce, fused Adam.  Weak scaling: 4096 rays per rank.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C_SDF, C_NERF, C_APP = 524544, 604160, 1211648          # MACs per point
PEAK_F32_MFMA = 157.3e12                                 # MI355X dense fp32 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_BF16_MFMA = 2500e12                                 # MI355X dense bf16 MFMA peak (same guide)
# fp32-equivalent ceilings of the arithmetic the dense layers run on (nero_amd/chain.py GEMM_MODE): the f32-input MFMA itself,
# the bf16 matrix pipe issuing 6 plane products per fp32 multiply-add (mlp_split.hip), the fp16 pipe issuing 3 (mlp_f16x3.hip)
PEAK_OF_MODE = {0: PEAK_F32_MFMA, 1: PEAK_BF16_MFMA / 6, 2: PEAK_BF16_MFMA / 3}
MFMA_OF_MODE = {0: 'v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain)',
                1: 'v_mfma_f32_32x32x16_bf16, 3 exact bf16 planes per operand, 6 plane products per fp32 multiply-add: peak = 2500 / 6',
                2: 'v_mfma_f32_32x32x16_f16, 2 block-scaled fp16 planes per operand, 3 plane products per fp32 multiply-add: peak = 2500 / 3'}


def hbm_traffic_per_launch(kernel):
    """average HBM bytes per launch of `kernel` from the committed PMC summary (scripts/prof_traffic.sh), or None"""
    path = os.path.join(ROOT, 'profiles', 'r01_hbm_traffic_per_kernel.csv')
    try:
        for line in open(path):
            f = line.strip().split(',')
            if len(f) == 5 and f[0] == kernel:
                return int((float(f[3]) + float(f[4])) * 1024)
    except OSError:
        pass
    return None


def cpu_baseline(cfg, variance, step, rays=512, budget_s=15.0, max_steps=10):
    """the oracle (a port of the reference's torch path, oracle/nero_oracle.py) timed on this box's host cores on a bounded
    sample of the same workload: `rays` rays x (64+64+32) samples, forward + loss + backward, 1 step."""
    from oracle import nero_oracle as O
    from nero_amd.renderer import NeROShapeRenderer
    from nero_amd.synthetic import perturb_state, synthetic_rays
    torch.manual_seed(6033)
    net = NeROShapeRenderer(cfg, training=False)
    perturb_state(net, variance)
    cores = min(os.cpu_count(), 32)            # beyond ~32 threads the small-matrix torch-CPU ops slow down
    torch.set_num_threads(cores)
    o, d, _, gt = synthetic_rays(rays, seed=1)
    sd = {k: v for k, v in net.named_parameters()}
    sd.update({k: v for k, v in net.named_buffers()})
    g = torch.Generator().manual_seed(3)
    rand1, rand_bg = torch.rand(rays, 1, generator=g), torch.rand(rays, 32, generator=g)
    c = {**O.DEFAULT_CFG, **cfg}
    near, far = O.near_far_from_sphere(o, d)
    t0, n = time.time(), 0
    while n < max_steps and (n == 0 or time.time() - t0 < budget_s):
        for p in sd.values():
            if p.grad is not None:
                p.grad = None
        P = O.effective_params(sd)
        out = O.render(P, c, o, d, near, far, torch.zeros(rays, 3, 4), step, O.anneal(c, step), rand1, rand_bg)
        loss = O.training_loss(c, out, gt, step)
        loss.backward()
        n += 1
    dt = time.time() - t0
    return {'value': rays * n / dt, 'unit': 'rays/s', 'cores': cores, 'kind': 'port',
            'sample': f'{rays} rays x (64+64+32) samples, oracle (torch-CPU port of the reference path) forward+loss+backward, '
                      f'{n} step(s), {dt:.1f} s'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--rays', type=int, default=4096)
    ap.add_argument('--train-step', type=int, default=25000, help='training-schedule step the batch is evaluated at')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    import torch.distributed as dist
    if world > 1:
        torch.cuda.set_device(local)
        dist.init_process_group('nccl')
    dev = f'cuda:{local}'
    torch.cuda.set_device(dev)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from nero_amd import _lib as L
    from nero_amd.train import ShapeTrainStep
    import ctypes as C

    cfg = {'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000}      # configs/shape/syn/bell.yaml
    variance = 0.5
    ts = ShapeTrainStep(cfg, rays_per_rank=args.rays, device=dev, variance=variance, rank=rank, world=world)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        info = ts.step(args.train_step + i)
    sync()
    t0 = time.time()
    n_in = n_out = 0
    for i in range(args.steps):
        info = ts.step(args.train_step + args.warmup + i)
        n_in += info['n_in']
        n_out += info['n_out']
    sync()
    dt = time.time() - t0
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt)
    rays_total = args.rays * world * args.steps
    value = rays_total / dt

    # ---- the same step on the exact-fp32 MFMA engine (NERO_GEMM=f32), reported next to the headline: 2 warmup + 5 timed steps --
    from nero_amd import chain as _CHM
    alt = None
    if _CHM.GEMM_MODE['fwd'] != L.GEMM_F32:
        _saved_modes = dict(_CHM.GEMM_MODE)
        _CHM.set_gemm_mode('f32')
        for i in range(2):
            ts.step(args.train_step + 200 + i)
        sync()
        t1 = time.time()
        for i in range(5):
            ts.step(args.train_step + 202 + i)
        sync()
        dta = torch.tensor([time.time() - t1], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(dta, op=dist.ReduceOp.MAX)
        alt = {'value': round(args.rays * world * 5 / float(dta), 1), 'unit': 'rays/s', 'ms_per_step': round(float(dta) / 5 * 1e3, 3),
               'steps': 5, 'mfma': 'v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain, 157.3 TFLOP/s peak)'}
        _CHM.GEMM_MODE.update(_saved_modes)
        ts.step(args.train_step + 300)           # back on the default engine before the roofline leg
        sync()

    # ---- forward-only (inference) rays/s on the default engines: 2 warmup + 5 timed renders --------------------
    for i in range(2):
        ts.forward_only(args.train_step + 400 + i)
    sync()
    t2 = time.time()
    for i in range(5):
        ts.forward_only(args.train_step + 402 + i)
    sync()
    dtf = torch.tensor([time.time() - t2], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(dtf, op=dist.ReduceOp.MAX)
    fwd_only = {'value': round(args.rays * world * 5 / float(dtf), 1), 'unit': 'rays/s', 'ms_per_render': round(float(dtf) / 5 * 1e3, 3),
                'what': "renderer.render(is_train=False): sampler + render forward + the reference's validation extras, no loss / backward"}

    # ---- roofline leg: per-launch HIP-event timing of the MFMA kernel classes, on extra (untimed) steps ------------
    roof = None
    if rank == 0:
        L.lib.nero_prof_enable(1)
        for i in range(3):
            ts.step(args.train_step + 100 + i)
        torch.cuda.synchronize()
        L.lib.nero_prof_enable(0)
        rep = (C.c_double * 12)()
        L.lib.nero_prof_report(rep)
        from nero_amd import chain as CH
        kname = {'fwd': ('mlp_fwd_kernel', 'fwd_split_kernel', 'fwd_f16_kernel'), 'tan': ('mlp_tan_kernel', 'tan_split_kernel', 'tan_f16_kernel'),
                 'bwd': ('mlp_bwd_kernel', 'bwd_split_kernel', 'bwd_f16_kernel'), 'dw': ('dw_gemm_kernel', 'dw_split_kernel', 'dw_split_kernel')}
        passes = ('fwd', 'tan', 'bwd', 'dw')
        kinds = [kname[m][CH.GEMM_MODE[m]] for m in passes]
        peaks = [PEAK_OF_MODE[CH.GEMM_MODE[m]] for m in passes]
        rows = [(kinds[k], rep[3 * k], rep[3 * k + 1], rep[3 * k + 2], peaks[k], CH.GEMM_MODE[passes[k]]) for k in range(4)]
        dom = max(rows, key=lambda r: r[2])
        ach = dom[3] / (dom[2] * 1e-3) / 1e12 if dom[2] > 0 else 0.0
        peak = dom[4] / 1e12
        roof = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                'frac': round(ach / peak, 4), 'traffic': hbm_traffic_per_launch(dom[0]), 'kernel': dom[0], 'mfma': MFMA_OF_MODE[dom[5]],
                'traffic_source': 'profiles/r01_hbm_traffic_per_kernel.csv: rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE, '
                                  'separate passes, bytes per launch (null if the kernel is not in that file)',
                'avg_launch_ms': round(dom[2] / max(dom[1], 1), 4),
                'per_kernel': {r[0]: {'launches_per_step': r[1] / 3, 'ms_per_step': round(r[2] / 3, 3),
                                      'tflops': round(r[3] / (r[2] * 1e-3) / 1e12, 2) if r[2] > 0 else 0.0,
                                      'peak': round(r[4] / 1e12, 1)} for r in rows}}
    elif world > 1:
        for i in range(3):
            ts.step(args.train_step + 100 + i)

    if rank == 0:
        from nero_amd import chain as _CH
        CH_SPLIT = _CH.GEMM_MODE['fwd'] != L.GEMM_F32
        CH_MODES = {k: {0: 'f32', 1: 'bf16x6', 2: 'f16x3'}[v] for k, v in _CH.GEMM_MODE.items()}
        # whole-step algorithmic FLOPs with the measured inner/outer split of this rank
        sampler_evals = args.rays * (64 + 3 * 16)
        flop_step = (n_in / args.steps) * 2 * (6 * C_SDF + 3 * C_APP) + (n_out / args.steps) * 2 * 3 * C_NERF + sampler_evals * 2 * C_SDF
        res = {
            'metric': 'training rays/sec (Stage-I shape, 128 samples/ray)', 'value': round(value, 1), 'unit': 'rays/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': ('f32 (dense layers: fp32 operands carried as exact / block-scaled 16-bit plane pairs or triples on the bf16/fp16 '
                      f'matrix pipe with fp32 accumulation, fp32-grade error; modes {CH_MODES})') if CH_SPLIT else 'f32',
            'data': 'synthetic',
            'config': {'workload': "GlossySynthetic 'bell' Stage-I shape, 4096 rays x (64+64+32) samples per GPU, "
                                   f'training-schedule step {args.train_step}', 'rays_per_gpu': args.rays,
                       'parallelism': f'dp{world}', 'optimizer': 'adam(fused)', 'inv_s': 'exp(10*0.5)'},
            'step_mlp_flop_frac': round(flop_step / (dt / args.steps) / PEAK_OF_MODE[_CH.GEMM_MODE['fwd']], 4),
            'step_mlp_flop_frac_of_f32_mfma_peak': round(flop_step / (dt / args.steps) / PEAK_F32_MFMA, 4),
            'inner_samples_per_ray': round(n_in / args.steps / args.rays, 2),
            'forward_only': fwd_only,
            'roofline': roof,
        }
        if alt is not None:
            res['f32_mfma_engine'] = alt
        if not args.no_cpu_baseline and world == 1:          # (rank 0 at N = 1 only)
            res['cpu_baseline'] = cpu_baseline(cfg, variance, args.train_step)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
