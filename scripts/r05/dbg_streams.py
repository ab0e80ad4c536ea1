"""Run-to-run comparison of the Stage-I training step, tensor by tensor and intermediate by intermediate (round 5).
usage: [NERO_STREAMS=3 NERO_DW_JOIN=e|j|t|b|l] python scripts/r05/dbg_streams.py [bell|bear] [rays]
Ten repeats of the same batch with the same draws; prints which gradient tensors, which of nine intermediates of the backward
(nero_stage1_debug_buffers), which pieces of the forward state and which glue buffers differ from the first repeat, and for d_grad the rows.
This is the script that traced the 3-stream nondeterminism to sdf_alpha_bwd's packed fp32 instructions: with a library built
WITH them (scripts/build_variant.sh slp shade -fslp-vectorize; NERO_HIP_LIB=...) two of nine repeats differ in 16-row blocks of d_grad."""
import os, sys
sys.path.insert(0, '.')
import torch
from nero_amd.train import ShapeTrainStep
kind = sys.argv[1] if len(sys.argv) > 1 else 'bear'
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 512
BELL = {'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000}
cfg = dict(BELL) if kind == 'bell' else {**BELL, 'shader_config': {'human_light': True}}
ts = ShapeTrainStep(cfg, rays_per_rank=rays, pool_rays=4 * rays, device='cuda:0', variance=0.5, prime_fraction=0.0, prime_passes=0)
names = ts.fopt.names + ['variance']
import ctypes as C
from nero_amd import stage1 as S1
LBL = ['d_geo', 'd_feat', 'd_sdf4', 'd_grad', 'dinv', 'ehat', 'adot', 'd_alpha_inner', 'd_metallic_raw']


def snapshot():
    ptrs, nb = (C.c_void_p * 9)(), (C.c_size_t * 9)()
    S1._lib.nero_stage1_debug_buffers(ts.drv.h, ptrs, nb)
    out = {}
    ws = ts.drv.workspace(rays)
    base = ws.data_ptr()
    for l, p_, b in zip(LBL, ptrs, nb):
        if p_ and b:
            off = p_ - base
            out[l] = ws[off:off + b].clone()                              # (raw bytes: bit comparison)
    st = ts.drv.state()                                                   # forward state the backward reads
    n_in = st.n_in
    rp = (n_in + 63) // 64 * 64
    for l, p_, b in (('fwd.normal', st.normal, n_in * 12), ('fwd.sdf4', st.sdf4, rp * 16), ('fwd.x4', st.x4, rp * 16), ('fwd.geo', st.geo, rp * 32),
                     ('fwd.inner_idx', st.inner_idx, n_in * 4)):
        off = p_ - base
        out[l] = ws[off:off + b].clone()
    g = ts._glue_obj
    if g is not None:
        B = g._bufs[rays]
        for l in ('d_gerr', 'd_rgb', 'gerr', 'rgb'):
            out['glue.' + l] = B[l].clone().view(torch.uint8).reshape(-1)
    return out


ref = None
ref_snap = None
side = torch.cuda.Stream() if os.environ.get('NERO_DBG_SIDE_STREAM') else None
for k in range(10):
    ts.cursor = 0
    torch.manual_seed(1234)
    if side is not None:
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            info = ts.forward_backward(25000)
    else:
        info = ts.forward_backward(25000)
    torch.cuda.synchronize()
    cur = ts.bucket.flat.clone()
    snap = snapshot()
    if ref is None:
        ref, ref_snap = cur, snap
        continue
    dsnap = [(l, int((snap[l] != ref_snap[l]).sum()), snap[l].numel()) for l in snap if not torch.equal(snap[l], ref_snap[l])]
    if dsnap:
        extra = ''
        if 'd_grad_copy' in snap:
            extra = (f" | within THIS repeat: d_grad_copy == d_grad: {torch.equal(snap['d_grad_copy'], snap['d_grad'][:snap['d_grad_copy'].numel()])}, "
                     f"d_geo_copy == d_geo: {torch.equal(snap['d_geo_copy'], snap['d_geo'])}")
        print(f'repeat {k}: differing intermediates: {dsnap}{extra}', flush=True)
        a, b = snap['d_grad'].view(torch.float32), ref_snap['d_grad'].view(torch.float32)
        ix = torch.nonzero(a != b)[:, 0]
        rows = torch.unique(ix // 3)
        print(f'   d_grad: {ix.numel()} floats in {rows.numel()} rows; rows: {rows[:60].tolist()}', flush=True)
        r0 = int(rows[0])
        print(f'   row {r0}: now {a[3 * r0:3 * r0 + 3].tolist()} ref {b[3 * r0:3 * r0 + 3].tolist()}', flush=True)
        g = snap['d_geo'].view(torch.float32).reshape(-1, 8)
        print(f'   d_geo row {r0}: {g[r0].tolist()}', flush=True)
        if k > 3:
            break
    if not torch.equal(cur, ref):
        off = 0
        bad = []
        for n, p in zip(names, ts.bucket.params):
            a, b = cur[off:off + p.numel()], ref[off:off + p.numel()]
            off += p.numel()
            if not torch.equal(a, b):
                bad.append((n, int((a != b).sum()), p.numel(), float((a - b).abs().max() / (b.abs().max() + 1e-30))))
        print(f'repeat {k}: streams={os.environ.get("NERO_STREAMS")} differing tensors:', bad, flush=True)
    else:
        print(f'repeat {k}: identical', flush=True)
