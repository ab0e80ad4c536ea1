"""bench.py -- Stage-I training rays/sec on MI355X (BASELINE.json metric).  One JSON line on rank 0.

  python bench.py --gpus N --steps K --warmup W
        N = 1: runs in this process.  N > 1 without a torch.distributed environment: bench.py re-launches itself as
        `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`
        (one rank per GPU over RCCL) and forwards the one JSON line; launched that way by the driver it just runs as a rank.

A step = one full training step of the GlossySynthetic 'bell' Stage-I shape configuration (configs[1]): 4096 rays per GPU x
(64 coarse + 64 importance + 32 background) samples, hierarchical sampling (112 no-grad SDF evals/ray), render forward, loss
(incl. the occlusion loss: schedule step 25000), backward including the second-order SDF term, flat RCCL gradient all-reduce,
fused Adam.  Weak scaling: 4096 rays per rank.  Inputs are resident in HBM (device ray pool).

Besides the contract fields the line carries: `roofline` (dominant MFMA kernel, live HIP-event timing), `protocol_8d` (SURVEY.md
8d: median of >= 50 steps after >= 10 warm-up), `forward_only`, `f32_mfma_engine`, `stage2` (BASELINE configs[3]: P = 4096 surface
points x 128+128 and the YAML default 512+256 MC directions), `reference_gpu_baseline` + `x_reference_gpu` (the UNMODIFIED reference
-- oracle/_ref/, packaged by oracle/make_ref.py -- timed on the same MI355X by oracle/run_ref.py: the ">= 10x reference single-GPU
PyTorch" yard-stick of north_star), `torch_gpu_baseline` (our torch port of the same algorithm, same GPU), and `cpu_baseline` (kind
"reference": the unmodified reference on this box's host cores, the port beside it).  The baseline legs are the only places that
touch oracle/.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
# The split-sum FG table is INPUT DATA of the reference (assets/bsdf_256_256.bin, read relative to the working directory at
# network/field.py:510).  The bench does not run from a reference checkout, so it names the table explicitly -- the committed fixture that
# holds the reference's asset bit for bit (tests/test_fg_lut.py::test_constructed_in_the_reference_tree) -- instead of letting the model fall
# back, loudly, to the computed table (within 3.9e-4 of the asset: VERDICT r5 missing 6).  An explicit $NERO_FG_LUT wins.
_FG_FIXTURE = os.path.join(ROOT, 'tests', 'golden', 'fg_lut_ref.npz')
if 'NERO_FG_LUT' not in os.environ and os.path.exists(_FG_FIXTURE):
    os.environ['NERO_FG_LUT'] = _FG_FIXTURE
sys.path.insert(0, ROOT)

C_SDF, C_NERF, C_APP = 524544, 604160, 1211648          # MACs per point
C_MAT, C_OUTER, C_INNER = 1078272, 150272, 163328       # Stage II
PEAK_F32_MFMA = 157.3e12                                 # MI355X dense fp32 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_BF16_MFMA = 2500e12                                 # MI355X dense bf16 / fp16 MFMA peak (same guide)
# fp32-equivalent ceilings of the arithmetic the dense layers run on (nero_amd/chain.py GEMM_MODE): the f32-input MFMA itself,
# the bf16 matrix pipe issuing 6 plane products per fp32 multiply-add (mlp_split.hip), the fp16 pipe issuing 3 (mlp_f16x3.hip)
PEAK_OF_MODE = {0: PEAK_F32_MFMA, 1: PEAK_BF16_MFMA / 6, 2: PEAK_BF16_MFMA / 3, 3: PEAK_BF16_MFMA / 3}
MFMA_OF_MODE = {0: 'v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain)',
                1: 'v_mfma_f32_32x32x16_bf16, 3 exact bf16 planes per operand, 6 plane products per fp32 multiply-add: peak = 2500 / 6',
                2: 'v_mfma_f32_32x32x16_f16, 2 block-scaled fp16 planes per operand, 3 plane products per fp32 multiply-add: peak = 2500 / 3'}
MFMA_OF_MODE[3] = MFMA_OF_MODE[2]
BELL = {'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000}      # configs/shape/syn/bell.yaml
VARIANCE = 0.5


def lib_sha12():
    import hashlib
    try:
        return hashlib.sha256(open(os.path.join(ROOT, 'nero_amd', 'libnero_hip.so'), 'rb').read()).hexdigest()[:12]
    except OSError:
        return None


def hbm_traffic_per_launch(kernel, prefix='hbm_traffic_per_kernel'):
    """average HBM bytes per launch of `kernel` from the newest committed PMC summary (scripts/prof_traffic.sh), or None.
    -> (bytes, description of the source incl. the commit / library hash the counters were taken on and whether that library is
    the one running now)"""
    for rnd in ('r06', 'r05', 'r04', 'r03', 'r02', 'r01'):
        name = f'{rnd}_{prefix}.csv'
        try:
            lines = open(os.path.join(ROOT, 'profiles', name)).read().splitlines()
        except OSError:
            continue
        stamp = next((ln[1:].strip() for ln in lines if ln.startswith('# taken on')), 'taken on: not recorded (round <= 2 summary)')
        for line in lines:
            f = line.strip().split(',')
            # (round 4 launches the weight-gradient jobs in groups: `dw_f16_kernel`'s counters are under `dw_f16_batch_kernel`)
            if len(f) in (5, 6) and f[0] in (kernel, kernel.replace('_kernel', '_batch_kernel')):      # kernel, launches, fetch_kb_raw, fetch_kb (x2 corrected), write_kb[, gb_per_step]
                cur = lib_sha12()
                same = (cur is not None and cur in stamp)
                return int((float(f[3]) + float(f[4])) * 1024), (f'profiles/{name}: rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE, '
                                                                   f'separate passes, bytes per launch; {stamp}; library running now: {cur} '
                                                                   f'({"same build" if same else "DIFFERENT build: counters may be stale"})')
    return None, None


def mfma_busy_of(kernel):
    """share of all SIMD-cycles of the launches of `kernel` (every template instance) in which the matrix pipe was occupied
    (profiles/r03_mfma_busy_per_kernel.csv: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch cycles), scripts/prof_mfma_busy.sh), or None"""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_mfma_busy_per_kernel.csv')))
    if not paths:
        return None
    busy = cyc = 0.0
    with open(paths[-1]) as f:
        for line in f:
            if line.startswith(kernel) or line.startswith(kernel.replace('_kernel', '_batch_kernel')):
                p = line.rstrip().rsplit(',', 6)
                try:
                    n, cycles, mb = float(p[1]), float(p[2]), float(p[3])
                except (ValueError, IndexError):
                    continue
                busy += n * mb
                cyc += n * cycles * 1024.0
    if cyc <= 0:
        return None
    # (the counters describe the build they were taken on: say whether that is the library running now, as for the traffic counters)
    with open(paths[-1]) as f:
        stamp = next((ln[1:].strip() for ln in f if ln.startswith('# taken on')), 'taken on: not recorded')
    cur = lib_sha12()
    same = cur is not None and cur in stamp
    return {'frac': round(busy / cyc, 4), 'source': os.path.relpath(paths[-1], ROOT) + ' (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES)',
            'taken_on': stamp, 'library_now': cur, 'same_build': bool(same)}


def cpu_baseline(cfg, variance, step, rays=512, budget_s=15.0, max_steps=10):
    """the oracle (a port of the reference's torch path, oracle/nero_oracle.py) timed on this box's host cores on a bounded
    sample of the same workload: `rays` rays x (64+64+32) samples, forward + loss + backward."""
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
    rand1, rand_bg, keys = torch.rand(rays, 1, generator=g), torch.rand(rays, 32, generator=g), torch.rand(rays * 160, generator=g)
    c = {**O.DEFAULT_CFG, **cfg}
    near, far = O.near_far_from_sphere(o, d)
    t0, n = time.time(), 0
    while n < max_steps and (n == 0 or time.time() - t0 < budget_s):
        for p in sd.values():
            if p.grad is not None:
                p.grad = None
        P = O.effective_params(sd)
        out = O.render(P, c, o, d, near, far, torch.zeros(rays, 3, 4), step, O.anneal(c, step), rand1, rand_bg, keys)
        loss = O.training_loss(c, out, gt, step)
        loss.backward()
        n += 1
    dt = time.time() - t0
    return {'value': rays * n / dt, 'unit': 'rays/s', 'cores': cores, 'kind': 'port',
            'sample': f'{rays} rays x (64+64+32) samples, oracle (torch-CPU port of the reference path) forward+loss+backward, '
                      f'{n} step(s), {dt:.1f} s'}


def run_reference(device, rays, samples, warmup, steps, threads=0, budget_s=0.0, timeout=900):
    """oracle/run_ref.py in a subprocess: the UNMODIFIED reference (oracle/_ref/, packaged by oracle/make_ref.py in the build container;
    it rides along to the GPU box like the built .so) timed on this box.  -> its JSON record, {'ok': False, 'error': ...} verbatim when it
    fails, or None when oracle/_ref/ is not there (then the baselines below fall back to the port and say so)."""
    ref_dir = os.path.join(ROOT, 'oracle', '_ref')
    if not os.path.exists(os.path.join(ref_dir, 'nero_ref.zip')):
        return None
    cmd = [sys.executable, os.path.join(ROOT, 'oracle', 'run_ref.py'), '--device', device, '--rays', str(rays), '--samples', *map(str, samples),
           '--warmup', str(warmup), '--steps', str(steps), '--variance', str(VARIANCE)]
    if threads:
        cmd += ['--threads', str(threads)]
    if budget_s:
        cmd += ['--budget-s', str(budget_s)]
    env = dict(os.environ, NERO_REFERENCE_ROOT=ref_dir)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'TORCHELASTIC_RUN_ID'):
        env.pop(k, None)
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return {'ok': False, 'error': f'timeout after {timeout} s', 'cmd': ' '.join(cmd[1:])}
    for line in reversed(p.stdout.strip().splitlines()):
        if line.startswith('{'):
            try:
                return json.loads(line)
            except ValueError:
                break
    return {'ok': False, 'error': 'no JSON line', 'returncode': p.returncode, 'stderr_tail': p.stderr[-1500:]}


def reference_cpu_record():
    """the newest committed profiles/rNN_ref_vs_port_cpu.json: the UNMODIFIED reference (shimmed, network/renderer.py:608-627) and the
    oracle port timed on the same host cores in the build container by oracle/ref_vs_port_cpu.py (BASELINE.md section 3's C1 protocol).
    Read-only bookkeeping: nothing under oracle/ or /root/reference is touched here."""
    paths = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_ref_vs_port_cpu.json')))
    if not paths:
        return None
    try:
        with open(paths[-1]) as f:
            r = json.load(f)
    except (OSError, ValueError):
        return None
    return {'value': r['reference']['rays_per_s'], 'unit': 'rays/s', 'cores': r['cores'], 'kind': 'reference', 'where': r['where'],
            'workload': r['workload'], 'protocol': r['protocol'], 'port_on_the_same_cores': r['port']['rays_per_s'],
            'port_over_reference': r['port_over_reference'], 'source': os.path.relpath(paths[-1], ROOT)}


def torch_gpu_baseline(cfg, variance, step, rays, dev, warmup=10, steps=50):
    """BASELINE.md section 3 'GPU reference denominator': the same algorithm (oracle/nero_oracle.py: per-layer ATen GEMMs through
    hipBLASLt, boolean-mask gathers, autograd double backward -- what the reference's PyTorch path does) on THIS MI355X, same
    workload, same schedule step, fused Adam; median of `steps` after `warmup`."""
    from oracle import nero_oracle as O
    from nero_amd.renderer import NeROShapeRenderer
    from nero_amd.synthetic import perturb_state, synthetic_rays
    torch.manual_seed(6033)
    net = NeROShapeRenderer(cfg, training=False)
    perturb_state(net, variance)
    net = net.to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)
    o, d, _, gt = synthetic_rays(rays * 4, seed=1)
    o, d, gt = o.to(dev), d.to(dev), gt.to(dev)
    c = {**O.DEFAULT_CFG, **cfg}
    zeros = torch.zeros(rays, 3, 4, device=dev)
    prev = torch.get_default_device()
    torch.set_default_device(dev)                       # the oracle creates its temporaries on the default device (like renderer.py:609)
    try:
        def one(i):
            s = slice((i % 4) * rays, (i % 4 + 1) * rays)
            opt.zero_grad(set_to_none=True)
            sd = {k: v for k, v in net.named_parameters()}
            sd.update({k: v for k, v in net.named_buffers()})
            P = O.effective_params(sd)
            near, far = O.near_far_from_sphere(o[s], d[s])
            out = O.render(P, c, o[s], d[s], near, far, zeros, step, O.anneal(c, step), torch.rand(rays, 1), torch.rand(rays, 32),
                           torch.rand(rays * 160))
            O.training_loss(c, out, gt[s], step).backward()
            opt.step()
        for i in range(warmup):
            one(i)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        torch.cuda.synchronize()
        ev[0].record()
        for i in range(steps):
            one(i)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
        peak = torch.cuda.max_memory_allocated() / 2 ** 30
    finally:
        torch.set_default_device(prev)
    med = ms[len(ms) // 2]
    return {'value': round(rays / (med * 1e-3), 1), 'unit': 'rays/s', 'ms_per_step_median': round(med, 2), 'ms_per_step_min': round(ms[0], 2),
            'ms_per_step_max': round(ms[-1], 2), 'warmup': warmup, 'steps': steps, 'kind': 'port',
            'what': f'oracle/nero_oracle.py (torch port of the reference path) on the same MI355X, {rays} rays x (64+64+32), schedule step '
                    f'{step}, fp32 ATen ops + fused Adam, peak {peak:.1f} GiB'}


def _bench_mesh(subdiv=7):
    import numpy as np
    from nero_amd.synthetic import icosphere
    v, f = icosphere(subdiv, 0.5, 0.2)
    return v, np.ascontiguousarray(f[:, ::-1])


STAGE2_CFG = {'bell': dict(human_lights=False, outer_light_version='direction'),                 # configs/material/syn/bell.yaml
              'bear': dict(human_lights=True, outer_light_version='sphere_direction')}            # configs/material/real/bear.yaml


def stage2_step_bench(dev, kind, P_, Dd, Ds, mesh, rank=0, world=1, warmup=5, steps=20, sync=None, max_over_ranks=None, prof=True):
    """K Stage-II training steps through nero_amd.train.MaterialTrainStep (fused weight-norm / Adam kernels, flat gradient bucket,
    RCCL all-reduce when world > 1) on P_ surface points per rank x (Dd + Ds) MC directions: materials, direction sampling, BVH trace
    of the P*D secondary rays, hit / miss light MLPs, microfacet estimator, regularisers, backward, optimiser.  pts/s, light-rays/s,
    tracer rays/s, the MLP-FLOP fraction of SURVEY.md 8d's model 3 [2 C_mat + D ((1-h) C_miss + h C_inner)] MACs per point with the
    measured hit fraction h, and (rank 0) the per-class MFMA kernel timing of three extra steps (`roofline`)."""
    import ctypes as C
    from nero_amd import _lib as L
    from nero_amd import chain as CH
    from nero_amd.train import MaterialTrainStep
    sync = sync or torch.cuda.synchronize
    mx = max_over_ranks or (lambda x: x)
    scfg = dict(diffuse_sample_num=Dd, specular_sample_num=Ds, **STAGE2_CFG[kind])
    ts = MaterialTrainStep({'shader_cfg': scfg, 'database_name': 'real/bear/raw_1024' if kind == 'bear' else 'syn/bell'}, mesh,
                           points_per_rank=P_, pool_points=4 * P_ * world, device=dev, rank=rank, world=world)
    tracer = ts.net.ray_tracer
    rec = {'n': 0, 'hit': 0, 'ms': 0.0, 'on': False}

    class Timed:                                            # times the tracer call of the step on the secondary rays of one untimed extra step
        def trace_grouped(self, ro, rd, group, heavy_from):
            return self.trace(ro, rd, (group, heavy_from))

        def trace_masked(self, ro, rd, skip, chunk_order=None):     # (round 6: the flags of the step's zero-weight rays -- not traversed -- and its launch order)
            return self.trace(ro, rd, None, skip, chunk_order)

        def trace(self, ro, rd, order=None, skip=None, chunk_order=None):
            if skip is not None or chunk_order is not None:
                run = lambda: tracer.trace_masked(ro, rd, skip, chunk_order=chunk_order)
            else:
                run = (lambda: tracer.trace_grouped(ro, rd, *order)) if order is not None else (lambda: tracer.trace(ro, rd))
            if not rec['on']:
                return run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = run()
            e1.record()
            torch.cuda.synchronize()
            rec['ms'] += e0.elapsed_time(e1)
            rec['n'] += ro.shape[0]
            rec['hit'] += int((r[2] < 10).sum())
            return r
    ts.net.ray_tracer = Timed()
    for i in range(warmup):
        ts.step(5000 + i)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    sync()
    t0 = time.time()
    ev[0].record()
    for i in range(steps):
        ts.step(5000 + warmup + i)
        ev[i + 1].record()
    sync()
    wall_local = (time.time() - t0) / steps
    wall = mx(wall_local * steps) / steps
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    rec['on'] = True
    ts.step(6000)
    rec['on'] = False
    torch.cuda.synchronize()
    D = Dd + Ds
    h = rec['hit'] / max(rec['n'], 1)
    # rows of the light MLPs per ray: the step's own counts (round 6: rays whose estimator weight is exactly zero -- GGX directions below the
    # shading horizon under the Schlick geometry term, nero_mc_dead_rays -- are neither traced nor shaded; NERO_MC_SKIP_DEAD=0 keeps them)
    n_miss_rows, n_hit_rows, n_rays, n_hum_rows = getattr(ts.drv, 'last_counts', None) or (round((1 - h) * rec['n']), rec['hit'], max(rec['n'], 1), None)
    m_frac, h_frac = n_miss_rows / n_rays, n_hit_rows / n_rays
    # (bear: the human-light MLP owns a row only for the miss rays that reach the photographer's region of the camera plane -- its output is
    #  multiplied by that mask, network/field.py:829; NERO_MC_SKIP_DEAD=0: for every miss ray, as rounds 1-5)
    u_frac = (n_hum_rows / n_rays) if (kind == 'bear' and n_hum_rows is not None) else (m_frac if kind == 'bear' else 0.0)
    c_outer = 168704 if kind == 'bear' else C_OUTER                                             # C_outer, C_human
    flop_pt = 2 * 3 * (2 * C_MAT + D * (m_frac * c_outer + u_frac * 138240 + h_frac * C_INNER))
    peak = PEAK_OF_MODE[CH.GEMM_MODE['fwd']]
    out = {'model': kind, 'points_per_gpu': P_, 'directions': f'{Dd}+{Ds}', 'n_gpus': world, 'trainer': 'fused (nero_wn_forward_batch / nero_wn_adam_batch)',
           'ms_per_step': round(wall * 1e3, 3), 'ms_per_step_this_rank': round(wall_local * 1e3, 3), 'ms_per_step_median': round(ms[len(ms) // 2], 3),
           'points_per_s': round(P_ * world / wall, 1), 'light_rays_per_s': round(P_ * world * D / wall, 1),
           'tracer_rays_per_s': round(rec['n'] / (rec['ms'] * 1e-3), 1) if rec['ms'] > 0 else None, 'tracer_ms': round(rec['ms'], 3),
           'hit_fraction': round(h_frac, 4), 'miss_fraction': round(m_frac, 4), 'zero_weight_ray_fraction': round(1.0 - m_frac - h_frac, 4), 'human_light_row_fraction': round(u_frac, 4),
           'mlp_mflop_per_point': round(flop_pt / 1e6, 1),
           'mlp_flop_frac': round(flop_pt * P_ / wall / peak, 4), 'mlp_flop_frac_peak_tflops': round(peak / 1e12, 1),
           'warmup': warmup, 'steps': steps}
    if prof and rank == 0:
        L.lib.nero_prof_enable(1)
        for i in range(3):
            ts.step(7000 + i)
        torch.cuda.synchronize()
        L.lib.nero_prof_enable(0)
        rep = (C.c_double * 24)()
        L.lib.nero_prof_report_kernels(rep)             # per KERNEL: 0-3 the 512-thread chain kernels + the weight-gradient GEMM, 4-6 mlp_f16p.hip's
        names = ('fwd_f16_kernel', 'tan_f16_kernel', 'bwd_f16_kernel', 'dw_f16_kernel', 'fwd_p_kernel', 'tan_p_kernel', 'bwd_p_kernel')
        rows = {names[k]: (rep[3 * k], rep[3 * k + 1], rep[3 * k + 2]) for k in range(7) if rep[3 * k] > 0}
        if rows:
            dom = max(rows, key=lambda k: rows[k][1])
            n_l, ms_l, fl = rows[dom]
            kern = dom
            traffic, tsrc = hbm_traffic_per_launch(kern, 'stage2_hbm_traffic_per_kernel')
            out['roofline'] = {'bound': 'mfma', 'kernel': kern, 'achieved': round(fl / (ms_l * 1e-3) / 1e12, 2), 'peak': round(peak / 1e12, 1),
                               'unit': 'TFLOP/s', 'frac': round(fl / (ms_l * 1e-3) / peak, 4), 'avg_launch_ms': round(ms_l / max(n_l, 1), 4),
                               'traffic': traffic, 'traffic_source': tsrc,
                               'per_kernel': {k: {'launches_per_step': v[0] / 3, 'ms_per_step': round(v[1] / 3, 3),
                                                  'tflops': round(v[2] / (v[1] * 1e-3) / 1e12, 2)} for k, v in rows.items()}}
    elif prof:
        for i in range(3):
            ts.step(7000 + i)
    del ts
    torch.cuda.empty_cache()
    return out


def stage2_bench(dev, P_=4096, configs=((128, 128), (512, 256)), warmup=5, steps=20, subdiv=7):
    """BASELINE configs[3]: bell Stage-II material step at P = 4096 x (128+128) and x (512+256, the YAML default)
    on a 327 680-triangle bumpy icosphere (stand-in for the extracted shape mesh)."""
    mesh = _bench_mesh(subdiv)
    out = {'mesh_triangles': int(mesh[1].shape[0]), 'points': P_, 'configs': []}
    for Dd, Ds in configs:
        out['configs'].append(stage2_step_bench(dev, 'bell', P_, Dd, Ds, mesh, warmup=warmup, steps=steps))
    return out


def inference_bench(dev, cfg, variance, res_grid=256):
    """SURVEY.md 8f rows 1-3 (the callers either side of the training path), forward only, on rank 0 at N = 1:
    mesh extraction (`extract_fields`, network/field.py:1090-1108 via extract_mesh.py:24-27), novel-view / validation rendering
    (`nvs` / `test_step`, network/renderer.py:189-222,274-317) at the reference's test_ray_num = 1024 and at 8192 rays per chunk,
    and the Stage-II dataset pre-trace (`_construct_ray_batch`, network/renderer.py:756-802) through the HIP BVH tracer."""
    import numpy as np
    from nero_amd.renderer import NeROShapeRenderer, NeROMaterialRenderer
    from nero_amd.synthetic import look_at_pose, perturb_state
    out = {}
    torch.manual_seed(6033)
    net = NeROShapeRenderer(dict(cfg), training=False)
    perturb_state(net, variance)
    net = net.to(dev)
    net.extract_fields(resolution=64)                                        # warm-up (packing, allocator)
    torch.cuda.synchronize()
    t0 = time.time()
    net.extract_fields(resolution=res_grid)
    torch.cuda.synchronize()
    d = time.time() - t0
    n = res_grid ** 3
    out['extract_fields'] = {'grid': f'{res_grid}^3', 'sdf_evals_per_s': round(n / d, 1), 'seconds': round(d, 3),
                             'seconds_at_512^3_extrapolated': round(d * (512 ** 3) / n, 2),
                             'mlp_tflops': round(n * 2 * C_SDF / d / 1e12, 1), 'includes': 'grid generation, |p| >= 1 mask, D2H copy of the grid'}
    h = w = 400
    K = np.array([[500., 0, w / 2], [0, 500., h / 2], [0, 0, 1]], np.float32)
    pose = look_at_pose(np.array([0.0, -3.0, 0.5], np.float32))
    for chunk in (1024, 8192):
        net.render_image(pose, K, 64, 64, chunk=chunk)
        torch.cuda.synchronize()
        t0 = time.time()
        net.render_image(pose, K, h, w, chunk=chunk)
        torch.cuda.synchronize()
        d = time.time() - t0
        out[f'nvs_chunk{chunk}'] = {'image': f'{h}x{w}', 'rays_per_s': round(h * w / d, 1), 'seconds': round(d, 3),
                                    'seconds_at_800x800_extrapolated': round(d * 4, 2)}
    del net
    v, f = _bench_mesh()
    mat = NeROMaterialRenderer({'database_name': 'synthetic'}, is_train=False, mesh=(v, f)).to(dev)
    imn, hh, ww = 16, 800, 800
    Ks = torch.tensor([[1000., 0, ww / 2], [0, 1000., hh / 2], [0, 0, 1]]).repeat(imn, 1, 1)
    poses = torch.stack([torch.as_tensor(look_at_pose(np.array([3.0 * np.cos(a), 3.0 * np.sin(a), 0.6], np.float32))) for a in np.linspace(0, 5, imn)]).to(dev)
    mat._trace_views(Ks[:1], poses[:1], 64, 64, dev)
    torch.cuda.synchronize()
    t0 = time.time()
    r = mat._trace_views(Ks, poses, hh, ww, dev)
    torch.cuda.synchronize()
    d = time.time() - t0
    out['stage2_pretrace'] = {'views': f'{imn} x {hh}x{ww}', 'triangles': int(f.shape[0]), 'rays_per_s': round(imn * hh * ww / d, 1), 'seconds': round(d, 3),
                              'hit_fraction': round(float(r[-1].float().mean()), 3),
                              'seconds_for_128_views_extrapolated': round(d * 128 / imn, 2), 'includes': 'ray generation + BVH trace in 2^20-ray chunks, all on the device'}
    return out


def dropin_trainer_bench(dev, cfg, rays, variance, step0=25000, warmup=5, steps=20):
    """what INTEGRATION.md option A gives a user of the reference Trainer: NeROShapeRenderer.forward({'step': s}) (pool slicing,
    _process_ray_batch, render, loss_rgb: network/renderer.py:319-330) under a plain torch.optim.Adam loop with the reference's loss
    assembly (train/trainer.py:120-140) -- torch owns the parameters, the autograd graph and the optimiser; no fused optimiser, no flat
    bucket -- at `rays` = train_ray_num."""
    import numpy as np
    from nero_amd.renderer import NeROShapeRenderer
    from nero_amd.synthetic import look_at_pose, perturb_state
    from nero_amd.train import shape_training_loss, warm_up_cos_lr
    torch.manual_seed(6033)
    net = NeROShapeRenderer({**cfg, 'train_ray_num': rays}, training=False)
    perturb_state(net, variance)
    net = net.to(dev)
    rg = np.random.default_rng(0)
    n_img, res = 8, 512
    az, el = rg.uniform(0, 2 * np.pi, n_img), rg.uniform(0.15, 1.2, n_img)
    cams = np.stack([np.cos(az) * np.cos(el), np.sin(az) * np.cos(el), np.sin(el)], -1) * 3.0
    poses = torch.from_numpy(np.stack([look_at_pose(c) for c in cams], 0))
    K = torch.tensor([[700.0, 0, res / 2], [0, 700.0, res / 2], [0, 0, 1]]).repeat(n_img, 1, 1)
    imgs = torch.from_numpy(np.random.default_rng(2).uniform(0, 1, (n_img, res, res, 3)).astype(np.float32))
    net.set_ray_pool(imgs, K, poses, device=dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)

    def one(i):
        st = step0 + i
        for g in opt.param_groups:
            g['lr'] = warm_up_cos_lr(st)
        opt.zero_grad()
        out = net({'step': st})
        loss = out['loss_rgb'].mean() + (out['gradient_error'] * 0.1).mean()
        if 'loss_occ' in out:
            loss = loss + out['loss_occ'].mean()
        loss.backward()
        opt.step()
    for i in range(warmup):
        one(i)
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(steps):
        one(warmup + i)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    del net, opt
    torch.cuda.empty_cache()
    return {'value': round(rays / dt, 1), 'unit': 'rays/s', 'ms_per_step': round(dt * 1e3, 3), 'rays': rays, 'steps': steps, 'warmup': warmup,
            'what': "NeROShapeRenderer.forward({'step': s}) + the reference's loss assembly + torch.optim.Adam(fused=True): the drop-in path of "
                    'INTEGRATION.md option A (one batched weight-norm autograd node, the C step driver, the optimiser of torch)'}


def spawn_ranks(n):
    """`python bench.py --gpus N` outside a torch.distributed launcher: become the launcher"""
    if torch.cuda.device_count() < n:
        print(json.dumps({'error': f'--gpus {n} requested but this node exposes {torch.cuda.device_count()} GPU(s)'}), flush=True)
        return 2
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def run_stage2(args, dev, rank, world, sync, max_over_ranks, rank_table=None):
    """`--stage 2`: the Stage-II material step as the headline (BASELINE configs[3] at N = 1; configs[4] = `--config bear --points 2048
    --dirs 256+256` at N = 8).  Same contract: W untimed steps, K timed between barrier + synchronize, max over ranks."""
    Dd, Ds = (int(x) for x in args.dirs.split('+'))
    P_ = args.points or (2048 if args.config == 'bear' else 4096)
    mesh = _bench_mesh()
    r = stage2_step_bench(dev, args.config, P_, Dd, Ds, mesh, rank, world, args.warmup, args.steps, sync, max_over_ranks)
    ranks = rank_table(r['ms_per_step_this_rank']) if rank_table else {}
    if rank != 0:
        return
    res = {'metric': 'Stage-II material training surface points/sec', 'value': r['points_per_s'], 'unit': 'points/s', 'n_gpus': world,
           'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': r['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak',
           'vs_baseline': None, 'dtype': 'f32 (dense layers as two block-scaled fp16 planes, 3 MFMA products, fp32 accumulation)',
           'data': 'synthetic',
           'config': {'workload': f"Glossy{'Real' if args.config == 'bear' else 'Synthetic'} '{args.config}' Stage-II material, {P_} surface points x "
                                  f'({Dd}+{Ds}) MC light directions per GPU, {int(mesh[1].shape[0])}-triangle mesh, step 5000',
                      'points_per_gpu': P_, 'parallelism': f'dp{world}', 'optimizer': 'adam(fused)',
                      'zero_weight_rays': ('neither traced nor shaded: GGX directions below the shading horizon have an estimator weight of exactly 0.0 '
                                           'under the Schlick geometry term (network/field.py:892-903, 987) -- same outputs and gradients; '
                                           'NERO_MC_SKIP_DEAD=0 processes every ray' if os.environ.get('NERO_MC_SKIP_DEAD', '1') != '0'
                                           else 'processed like every other ray (NERO_MC_SKIP_DEAD=0)')},
           'stage2': r, **ranks}
    if 'roofline' in r:
        res['roofline'] = r['roofline']
    print(json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--rays', type=int, default=4096)
    ap.add_argument('--stage', type=int, default=1, choices=(1, 2), help='1: Stage-I shape step (the BASELINE metric); 2: Stage-II material step')
    ap.add_argument('--config', default='bell', choices=('bell', 'bear'), help="bell = GlossySynthetic (no human light); bear = GlossyReal "
                    '(human light; BASELINE configs[2] is `--config bear --rays 1024 --gpus 8`)')
    ap.add_argument('--points', type=int, default=0, help='--stage 2: surface points per GPU (default 4096 bell / 2048 bear)')
    ap.add_argument('--dirs', default='128+128', help='--stage 2: diffuse+specular MC directions (BASELINE configs[4]: 256+256)')
    ap.add_argument('--train-step', type=int, default=25000, help='training-schedule step the batch is evaluated at')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--quick', action='store_true', help='headline + roofline only (development runs)')
    ap.add_argument('--dp-configs', action='store_true', help='N > 1 only: after the headline, also run BASELINE configs[2] (bear Stage I, 8192 rays '
                    'per global batch) and configs[4] (bear Stage II, 16384 points x 256+256) as data-parallel jobs of this world size')
    args = ap.parse_args()

    if args.gpus > 1 and not all(k in os.environ for k in ('WORLD_SIZE', 'RANK', 'MASTER_ADDR', 'MASTER_PORT')):
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    import torch.distributed as dist
    # torch.distributed.run set the FULL rendezvous (a scheduler that merely exports WORLD_SIZE is not a launcher): RCCL is initialised even
    # for ONE rank, so that the N = 1 point of a scaling run goes through the same collectives
    under_launcher = all(k in os.environ for k in ('WORLD_SIZE', 'RANK', 'MASTER_ADDR', 'MASTER_PORT'))
    if not under_launcher:
        world, rank, local = 1, 0, 0
    if under_launcher:
        torch.cuda.set_device(local)
        dist.init_process_group('nccl')
        if world == 1:
            from nero_amd import parallel
            parallel.FORCE_COLLECTIVES = True
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

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t)

    def rank_table(ms_local):
        """one record per rank (device name / UUID / its own ms per step): the line then shows WHICH devices ran and how evenly"""
        pr = torch.cuda.get_device_properties(dev)
        me = {'rank': rank, 'local_rank': local, 'device': pr.name, 'uuid': str(getattr(pr, 'uuid', 'unknown')), 'host': socket.gethostname(),
              'ms_per_step': round(ms_local, 3)}
        if not (dist.is_available() and dist.is_initialized()):
            return {'rccl_ranks': 0, 'backend': None, 'ranks': [me]}
        table = [None] * world
        dist.all_gather_object(table, me)
        return {'rccl_ranks': world, 'backend': dist.get_backend(), 'ranks': table}

    if args.stage == 2:
        run_stage2(args, dev, rank, world, sync, max_over_ranks, rank_table)
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return

    cfg = dict(BELL) if args.config == 'bell' else {**BELL, 'shader_config': {'human_light': True}}     # configs/shape/{syn/bell,real/bear}.yaml
    c_app = C_APP if args.config == 'bell' else 1349888                                                 # + the human-light MLP
    ts = ShapeTrainStep(cfg, rays_per_rank=args.rays, device=dev, variance=VARIANCE, rank=rank, world=world)

    # ---- the contract: W untimed steps, then EXACTLY K steps between barrier + synchronize, max over ranks -----------------------
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
    dt_local = time.time() - t0
    dt = max_over_ranks(dt_local)
    value = args.rays * world * args.steps / dt
    ranks = rank_table(dt_local / args.steps * 1e3)

    # ---- SURVEY.md 8d protocol: median of >= 50 steps (HIP events at the step boundaries) after >= 10 warm-up steps ------------
    proto = None
    if not args.quick:
        for i in range(max(0, 10 - args.warmup - args.steps)):
            ts.step(args.train_step + 500 + i)
        n8 = 50
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n8 + 1)]
        sync()
        ev[0].record()
        for i in range(n8):
            ts.step(args.train_step + 600 + i)
            ev[i + 1].record()
        sync()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n8))
        med = max_over_ranks(ms[n8 // 2])
        proto = {'steps': n8, 'warmup': max(10, args.warmup + args.steps), 'ms_per_step_median': round(med, 3), 'ms_per_step_min': round(ms[0], 3),
                 'ms_per_step_max': round(ms[-1], 3), 'rays_per_s_at_median': round(args.rays * world / (med * 1e-3), 1)}

    # ---- the same step on the exact-fp32 MFMA engine (NERO_GEMM=f32), reported next to the headline: 2 warmup + 5 timed steps --
    from nero_amd import chain as CH
    alt = None
    if CH.GEMM_MODE['fwd'] != L.GEMM_F32 and not args.quick and world == 1:
        saved = dict(CH.GEMM_MODE)
        CH.set_gemm_mode('f32')
        for i in range(2):
            ts.step(args.train_step + 200 + i)
        sync()
        t1 = time.time()
        for i in range(5):
            ts.step(args.train_step + 202 + i)
        sync()
        dta = max_over_ranks(time.time() - t1)
        alt = {'value': round(args.rays * world * 5 / dta, 1), 'unit': 'rays/s', 'ms_per_step': round(dta / 5 * 1e3, 3),
               'steps': 5, 'mfma': 'v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain, 157.3 TFLOP/s peak)'}
        CH.GEMM_MODE.update(saved)
        ts.step(args.train_step + 300)           # back on the default engine before the roofline leg
        sync()

    # ---- forward-only (inference) rays/s on the default engines: 2 warmup + 5 timed renders --------------------
    fwd_only = None
    if not args.quick:
        for i in range(2):
            ts.forward_only(args.train_step + 400 + i)
        sync()
        t2 = time.time()
        for i in range(5):
            ts.forward_only(args.train_step + 402 + i)
        sync()
        dtf = max_over_ranks(time.time() - t2)
        fwd_only = {'value': round(args.rays * world * 5 / dtf, 1), 'unit': 'rays/s', 'ms_per_render': round(dtf / 5 * 1e3, 3),
                    'what': "renderer.render(is_train=False): sampler + render forward + the reference's validation extras, no loss / "
                            'backward; packed operand images cached across renders (weights unchanged)'}

    # ---- roofline leg: per-launch HIP-event timing of the MFMA kernel classes, on extra (untimed) steps ------------
    roof = None
    if rank == 0:
        L.lib.nero_prof_enable(1)
        for i in range(3):
            ts.step(args.train_step + 100 + i)
        torch.cuda.synchronize()
        L.lib.nero_prof_enable(0)
        rep = (C.c_double * 24)()
        L.lib.nero_prof_report_kernels(rep)             # per KERNEL: 0-3 the 512-thread chain kernels + the weight-gradient GEMM, 4-6 mlp_f16p.hip's
        kname = {'fwd': ('mlp_fwd_kernel', 'fwd_split_kernel', 'fwd_f16_kernel'),
                 'tan': ('mlp_tan_kernel', 'tan_split_kernel', 'tan_f16_kernel'),
                 'bwd': ('mlp_bwd_kernel', 'bwd_split_kernel', 'bwd_f16_kernel'),
                 'dw': ('dw_gemm_kernel', 'dw_split_kernel', 'dw_f16_kernel')}
        passes = ('fwd', 'tan', 'bwd', 'dw')
        kinds = [kname[m][CH.GEMM_MODE[m]] for m in passes] + ['fwd_p_kernel', 'tan_p_kernel', 'bwd_p_kernel']
        peaks = [PEAK_OF_MODE[CH.GEMM_MODE[m]] for m in passes + passes[:3]]
        modes_k = [CH.GEMM_MODE[m] for m in passes + passes[:3]]
        rows = [(kinds[k], rep[3 * k], rep[3 * k + 1], rep[3 * k + 2], peaks[k], modes_k[k]) for k in range(7) if rep[3 * k] > 0]
        dom = max(rows, key=lambda r: r[2])
        ach = dom[3] / (dom[2] * 1e-3) / 1e12 if dom[2] > 0 else 0.0
        peak = dom[4] / 1e12
        traffic, tsrc = hbm_traffic_per_launch(dom[0])
        roof = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                'frac': round(ach / peak, 4), 'traffic': traffic, 'kernel': dom[0], 'mfma': MFMA_OF_MODE[dom[5]],
                'traffic_source': tsrc,
                'mfma_busy': mfma_busy_of(dom[0]),
                'avg_launch_ms': round(dom[2] / max(dom[1], 1), 4),
                'per_kernel': {r[0]: {'launches_per_step': r[1] / 3, 'ms_per_step': round(r[2] / 3, 3),
                                      'tflops': round(r[3] / (r[2] * 1e-3) / 1e12, 2) if r[2] > 0 else 0.0,
                                      'peak': round(r[4] / 1e12, 1)} for r in rows}}
        # HBM side of every class: PMC bytes per launch (committed CSV) / the launch time measured here; the class is labelled by the
        # ceiling it sits closer to (the weight-gradient GEMM and the tangent chain stream 4-5 TB/s: HBM-shaped, not MFMA-shaped)
        for r in rows:
            pk = roof['per_kernel'][r[0]]
            tb, _ = hbm_traffic_per_launch(r[0])
            if tb and r[1] > 0 and r[2] > 0:
                tbs = tb / (r[2] * 1e-3 / r[1]) / 1e12
                pk['hbm_tb_per_s'] = round(tbs, 2)
                pk['hbm_frac_of_8tb_s'] = round(tbs / 8.0, 3)
                pk['bound'] = 'hbm' if tbs / 8.0 > pk['tflops'] / pk['peak'] else 'mfma'
            mb = mfma_busy_of(r[0])
            if mb:
                pk['mfma_busy'] = mb['frac']
    elif world > 1:
        for i in range(3):
            ts.step(args.train_step + 100 + i)

    res = None
    if rank == 0:
        split = CH.GEMM_MODE['fwd'] != L.GEMM_F32
        modes = {k: {0: 'f32', 1: 'bf16x6', 2: 'f16x3'}[v] for k, v in CH.GEMM_MODE.items()}
        # whole-step algorithmic FLOPs with the measured inner/outer split of this rank
        sampler_evals = args.rays * (64 + 3 * 16)
        flop_step = (n_in / args.steps) * 2 * (6 * C_SDF + 3 * c_app) + (n_out / args.steps) * 2 * 3 * C_NERF + sampler_evals * 2 * C_SDF
        res = {
            'metric': 'training rays/sec (Stage-I shape, 128 samples/ray)', 'value': round(value, 1), 'unit': 'rays/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': ('f32 (dense layers: fp32 operands carried as exact / block-scaled 16-bit plane pairs or triples on the bf16/fp16 '
                      f'matrix pipe with fp32 accumulation, fp32-grade error; modes {modes})') if split else 'f32',
            'data': 'synthetic',
            'config': {'workload': f"Glossy{'Synthetic' if args.config == 'bell' else 'Real'} '{args.config}' Stage-I shape, {args.rays} rays x "
                                   f'(64+64+32) samples per GPU, training-schedule step {args.train_step}', 'rays_per_gpu': args.rays,
                       'parallelism': f'dp{world}', 'optimizer': 'adam(fused)', 'inv_s': 'exp(10*0.5)',
                       'streams': int(os.environ.get('NERO_STREAMS', '3')),
                       'arithmetic': 'fp16 two-plane operands, 3 MFMA products into ONE fp32 accumulator; no packed fp32 VALU (DESIGN.md 9.3)',
                       'f16_paired_mask': CH.f16_paired(),
                       'fg_table': ('reference asset (' + os.path.relpath(os.environ['NERO_FG_LUT'], ROOT) + ')') if os.environ.get('NERO_FG_LUT') else 'computed fallback'},
            'step_mlp_flop_frac': round(flop_step / (dt / args.steps) / PEAK_OF_MODE[CH.GEMM_MODE['fwd']], 4),
            'step_mlp_flop_frac_of_f32_mfma_peak': round(flop_step / (dt / args.steps) / PEAK_F32_MFMA, 4),
            'inner_samples_per_ray': round(n_in / args.steps / args.rays, 2),
            'protocol_8d': proto, 'forward_only': fwd_only, 'roofline': roof, **ranks,
        }
        if alt is not None:
            res['f32_mfma_engine'] = alt
    del ts
    import gc
    gc.unfreeze()
    gc.collect()
    torch.cuda.empty_cache()

    def leg(name, fn):
        """an extra measurement must never cost the headline line"""
        try:
            v = fn()
        except Exception as e:                                                # noqa: BLE001
            v = {'error': f'{type(e).__name__}: {e}'[:300]}
        if rank == 0 and v is not None:
            res[name] = v
        gc.collect()
        torch.cuda.empty_cache()

    if world > 1 and args.dp_configs and not args.quick:
        # (opt-in: an exception on ONE rank inside an extra collective leg would leave the others waiting in an all-reduce, and the
        # headline line of a scaling run must not depend on that)
        # BASELINE configs[2] and configs[4] as data-parallel jobs of THIS world size (every rank takes part; rank 0 reports):
        # bear Stage I with 8192 rays per global batch, bear Stage II with 16384 surface points x (256+256) directions
        def c2():
            r_ = max(64, 8192 // world)
            t2 = ShapeTrainStep({**BELL, 'shader_config': {'human_light': True}}, rays_per_rank=r_, device=dev, variance=VARIANCE, rank=rank,
                                world=world, prime_fraction=0.0)
            for i in range(5):
                t2.step(args.train_step + i)
            sync()
            t0_ = time.time()
            for i in range(20):
                t2.step(args.train_step + 5 + i)
            sync()
            d_ = max_over_ranks(time.time() - t0_) / 20
            return {'workload': f"GlossyReal 'bear' Stage-I shape, {r_ * world} rays per global batch = {r_} per GPU x {world} GPUs, RCCL flat "
                                'gradient all-reduce (BASELINE configs[2])', 'value': round(r_ * world / d_, 1), 'unit': 'rays/s',
                    'ms_per_step': round(d_ * 1e3, 3), 'rays_per_gpu': r_, 'n_gpus': world, 'steps': 20, 'warmup': 5}
        leg('configs2_bear_8192_rays_dp', c2)
        leg('configs4_bear_stage2_16384_points_dp',
            lambda: stage2_step_bench(dev, 'bear', max(64, 16384 // world), 256, 256, _bench_mesh(), rank, world, 5, 20, sync, max_over_ranks, prof=False))
    if rank == 0:
        if world == 1 and not args.quick:          # (baselines, the small-batch / drop-in legs and the Stage-II legs: rank 0 at N = 1 only)
            def r512():
                # (60 steps after 8: a 20-step sample of a 5 ms step is 0.1 s and scatters by +- 4 % from run to run -- scripts/r06/dropin_ab2.py)
                t5 = ShapeTrainStep(cfg, rays_per_rank=512, device=dev, variance=VARIANCE, prime_fraction=0.0)
                for i in range(8):
                    t5.step(args.train_step + i)
                torch.cuda.synchronize()
                t0_ = time.time()
                for i in range(60):
                    t5.step(args.train_step + 8 + i)
                torch.cuda.synchronize()
                d_ = (time.time() - t0_) / 60
                return {'value': round(512 / d_, 1), 'unit': 'rays/s', 'ms_per_step': round(d_ * 1e3, 3), 'rays': 512, 'steps': 60, 'warmup': 8,
                        'what': "the reference's own train_ray_num = 512 (configs/shape/syn/bell.yaml:31) on the fused training step"}
            leg('r512', r512)

            def c3_shard():
                # BASELINE configs[2] is bear Stage I, 8192 rays per global batch on 8 GPUs: THIS is one rank's shard of it (1024 rays, human light
                # on) on one GPU -- the N = 1 denominator of that configuration's scaling curve
                t3 = ShapeTrainStep({**BELL, 'shader_config': {'human_light': True}}, rays_per_rank=1024, device=dev, variance=VARIANCE, prime_fraction=0.0)
                for i in range(5):
                    t3.step(args.train_step + i)
                torch.cuda.synchronize()
                t0_ = time.time()
                for i in range(20):
                    t3.step(args.train_step + 5 + i)
                torch.cuda.synchronize()
                d_ = (time.time() - t0_) / 20
                return {'value': round(1024 / d_, 1), 'unit': 'rays/s', 'ms_per_step': round(d_ * 1e3, 3), 'rays': 1024, 'steps': 20, 'warmup': 5,
                        'what': "GlossyReal 'bear' Stage-I shape, 1024 rays x (64+64+32) on ONE GPU = one rank's shard of BASELINE configs[2] (8192 rays over 8 GPUs)"}
            leg('c3_shard_bear_1024_rays', c3_shard)
            leg('dropin_trainer', lambda: {'r4096': dropin_trainer_bench(dev, cfg, args.rays, VARIANCE, args.train_step),
                                           'r512': dropin_trainer_bench(dev, cfg, 512, VARIANCE, args.train_step, warmup=8, steps=60)})
            leg('inference', lambda: inference_bench(dev, cfg, VARIANCE))
            leg('stage2', lambda: stage2_bench(dev))
            # (= one rank's shard of BASELINE configs[4]: bear Stage II, 16384 points x (256+256) over 8 GPUs -- the N = 1 denominator)
            leg('stage2_bear_2048x512', lambda: stage2_step_bench(dev, 'bear', 2048, 256, 256, _bench_mesh()))
            tg = torch_gpu_baseline(cfg, VARIANCE, args.train_step, args.rays, dev)
            res['torch_gpu_baseline'] = tg
            # both baselines time the PORT (oracle/nero_oracle.py): the unmodified reference cannot travel to the GPU box.  The ratio below is
            # therefore "vs port"; the port / reference rate measured on the build container's host cores travels as a committed record
            # (profiles/rNN_ref_vs_port_cpu.json, oracle/ref_vs_port_cpu.py) and is reported BESIDE the ratio, never multiplied in.
            res['x_torch_gpu_baseline'] = round(res['protocol_8d']['rays_per_s_at_median'] / tg['value'], 2)
            res['x_torch_gpu_baseline_is'] = 'vs port (same algorithm as plain PyTorch ops on this GPU); see cpu_baseline.reference for port / reference'
            # ---- the UNMODIFIED reference on THIS box (oracle/_ref/, oracle/run_ref.py in a subprocess): GPU denominator of north_star's
            # ">= 10x the reference single-GPU PyTorch" (same C2 workload, >= 10 warm-up + 50 steps, median) and the host-core baseline
            rg = run_reference('cuda', args.rays, (64, 64, 32), 10, 50)
            if rg is not None:
                res['reference_gpu_baseline'] = rg
                if rg.get('ok'):
                    res['x_reference_gpu'] = round(res['protocol_8d']['rays_per_s_at_median'] / rg['rays_per_s'], 2)
                    res['x_reference_gpu_contract_loop'] = round(res['value'] / rg['rays_per_s'], 2)
                    res['x_reference_gpu_is'] = ('protocol_8d median rays/s of this framework / median rays/s of the unmodified reference (render + loss + '
                                                 'backward + torch Adam) on the same MI355X, same 4096 x (64+64+32) workload and schedule step')
            if not args.no_cpu_baseline:
                cores = min(os.cpu_count(), 32)
                port = cpu_baseline(cfg, VARIANCE, args.train_step)
                rc = run_reference('cpu', 512, (64, 64, 32), 1, 3, threads=cores, budget_s=25.0)
                if rc is not None and rc.get('ok'):
                    # ONE workload on ONE box: 512 rays x (64+64+32) of the benchmarked configuration on this box's host cores
                    res['cpu_baseline'] = {'value': rc['rays_per_s'], 'unit': 'rays/s', 'cores': rc['cores'], 'kind': 'reference',
                                           'sample': f"512 rays x (64+64+32) samples, schedule step {args.train_step}: the unmodified reference's render + loss + "
                                                     f"backward + Adam (oracle/run_ref.py on oracle/_ref), 1 warm-up + {rc['steps']} timed step(s), median "
                                                     f"{rc['s_per_step_median']:.2f} s/step",
                                           'port_on_the_same_cores': {'value': round(port['value'], 1), 'sample': port['sample']},
                                           'record': rc}
                    c1 = run_reference('cpu', 512, (32, 32, 32), 1, 3, threads=cores, budget_s=15.0)       # BASELINE.md section 3, configs[0]
                    if c1 is not None:
                        res['cpu_baseline']['c1_512x96'] = c1
                else:
                    res['cpu_baseline'] = port
                    res['cpu_baseline']['reference_unavailable'] = rc if rc is not None else 'oracle/_ref/ not present (python oracle/make_ref.py in the build container)'
                    refrec = reference_cpu_record()
                    if refrec:
                        res['cpu_baseline']['reference_build_container'] = refrec
        print(json.dumps(res), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
