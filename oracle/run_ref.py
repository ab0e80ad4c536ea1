"""Time the UNMODIFIED reference (packaged by oracle/make_ref.py into oracle/_ref/, or /root/reference where that exists) on this
box: its own NeROShapeRenderer.render + loss assembly + backward + torch.optim.Adam step -- what train/trainer.py:127-140 does per
iteration around network/renderer.py:608-627 -- on synthetic rays of the benchmark's shape.

MEASUREMENT INFRASTRUCTURE ONLY (bench.py's `cpu_baseline` kind "reference" and `reference_gpu_baseline` legs run this file in a
SUBPROCESS: the reference switches torch's default tensor type and the shim changes the working directory).  Nothing in nero_amd/
touches it.  Prints ONE JSON line.

    python oracle/run_ref.py --device cpu  --rays 512  --samples 64 64 32 --warmup 1  --steps 3  [--threads N]
    python oracle/run_ref.py --device cuda --rays 4096 --samples 64 64 32 --warmup 10 --steps 50

Third-party pieces the reference imports that are absent here are stubbed by oracle/ref_shim.py; the only one on the timed path is
nvdiffrast's `dr.texture` (network/field.py:612), restated there as a pure-torch bilinear fetch of the same table.  Everything else
executes the reference's own code, byte for byte (oracle/_ref/MANIFEST.json holds the hashes).
"""
import argparse
import json
import os
import sys
import time
import traceback

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--device', default='cpu', choices=('cpu', 'cuda'))
    ap.add_argument('--rays', type=int, default=512)
    ap.add_argument('--samples', type=int, nargs=3, default=(64, 64, 32), metavar=('N_SAMPLES', 'N_IMPORTANCE', 'N_BG'))
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--threads', type=int, default=0)
    ap.add_argument('--train-step', type=int, default=25000)
    ap.add_argument('--variance', type=float, default=0.5)
    ap.add_argument('--budget-s', type=float, default=0.0, help='stop timing early once this many seconds of timed steps have run (>= 1 step)')
    args = ap.parse_args()

    import torch
    from oracle import ref_shim
    from nero_amd.synthetic import perturb_state, synthetic_rays       # (pure torch / numpy helpers: no HIP library involved)
    cpu = args.device == 'cpu'
    if cpu:
        threads = args.threads or min(os.cpu_count(), 32)
        torch.set_num_threads(threads)
    renderer, _ = ref_shim.load_reference(force_cpu=cpu)
    ns, ni, nb = args.samples
    cfg = {'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000,          # configs/shape/syn/bell.yaml
           'n_samples': ns, 'n_importance': ni, 'n_bg_samples': nb}
    step = args.train_step
    torch.manual_seed(6033)
    net = renderer.NeROShapeRenderer(cfg, training=False)                # (training=False: no dataset on disk; the rays are handed to render())
    perturb_state(net, args.variance)
    net.train()
    R = args.rays
    o, d, _, gt = synthetic_rays(R * 4, seed=1)
    if not cpu:
        net = net.cuda()
        o, d, gt = o.cuda(), d.cuda(), gt.cuda()
        torch.set_default_tensor_type('torch.cuda.FloatTensor')          # network/renderer.py:609
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)                    # train/trainer.py:77-86 (optimizer_type adam)
    hp = torch.zeros(R, 3, 4)
    anneal = float(net.get_anneal_val(step))

    def one(i):
        s = slice((i % 4) * R, (i % 4 + 1) * R)
        opt.zero_grad()
        near, far = net.near_far_from_sphere(o[s], d[s])
        out = net.render(o[s], d[s], near, far, hp, -1, anneal, is_train=True, step=step)
        # train/trainer.py:131-137: the sum of the means of every 'loss*' entry (rgb loss, eikonal x 0.1, occlusion loss)
        loss = torch.mean(net.compute_rgb_loss(out['ray_rgb'], gt[s])) + torch.mean(out['gradient_error'] * 0.1) + torch.mean(out['loss_occ'])
        loss.backward()
        opt.step()
        return float(loss.detach()) if cpu else None

    for i in range(args.warmup):
        one(i)
    ts = []
    if cpu:
        t_all = time.time()
        for i in range(args.steps):
            t = time.time()
            one(args.warmup + i)
            ts.append(time.time() - t)
            if args.budget_s and time.time() - t_all > args.budget_s:
                break
        peak = None
    else:
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        ev[0].record()
        for i in range(args.steps):
            one(args.warmup + i)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = [ev[i].elapsed_time(ev[i + 1]) * 1e-3 for i in range(args.steps)]
        peak = round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)
    ts.sort()
    med = ts[len(ts) // 2]
    rec = {'ok': True, 'device': args.device, 'rays': R, 'samples': [ns, ni, nb], 'train_step': step, 'warmup': args.warmup, 'steps': len(ts),
           'rays_per_s': round(R / med, 2), 's_per_step_median': round(med, 5), 's_per_step_min': round(ts[0], 5), 's_per_step_max': round(ts[-1], 5),
           'rays_per_s_mean': round(R * len(ts) / sum(ts), 2),
           'cores': torch.get_num_threads() if cpu else None, 'peak_gib': peak, 'torch': torch.__version__,
           'reference_root': ref_shim.REF_ROOT,
           'what': 'unmodified reference: NeROShapeRenderer.render(is_train=True) + compute_rgb_loss + eikonal x 0.1 + occlusion loss, backward, '
                   'torch.optim.Adam.step (network/renderer.py:445-606, train/trainer.py:127-140); dr.texture = the pure-torch bilinear fetch '
                   'of oracle/ref_shim.py'}
    if not cpu:
        rec['gpu'] = torch.cuda.get_device_name(0)
    print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    try:
        main()
    except Exception as e:                                               # noqa: BLE001  (the caller records the failure verbatim)
        print(json.dumps({'ok': False, 'error': f'{type(e).__name__}: {e}', 'traceback': traceback.format_exc()[-1500:]}), flush=True)
        sys.exit(1)
