"""Host-side driver of the fused MLP-chain kernels (nero_amd/csrc/mlp_engine.hip): describes one network as a list of
entries, packs its effective weights into the MFMA operand images once per step, and launches forward / reverse /
weight-gradient passes through the C ABI (include/nero_hip.h).  No arithmetic happens here."""
import ctypes as C
import os
import math

import torch

from . import _lib as L




# arithmetic of the dense layers (include/nero_hip.h NERO_GEMM_*), all fp32-grade (tests/test_mlp_engine.py runs every test in
# every mode against the same fp64 reference and tolerance):
#   'f16x3'  (default) two block-scaled fp16 planes, 3 MFMA products: per activation row / weight matrix in the chain kernels
#            (mlp_f16x3.hip), per 16-row chunk with a running accumulator unit in the weight-gradient GEMM (mlp_f16dw.hip)
#   'bf16x6' three bf16 planes, 6 products (mlp_split.hip)
#            -- the chain passes of large launches run as two 256-thread workgroups per CU (mlp_f16p.hip; NERO_F16_PAIRED /
#            nero_f16_paired, default forward + tangent): an execution detail, same images, same results bit for bit.  (Rounds 2-3 had it
#            as a fourth arithmetic 'f16x3p', removed in round 4 for a sporadic wrong partial sum; round 5 found the cause -- packed fp32
#            beside MFMAs, DESIGN.md 9.3 -- and brought the kernels back; the MODE name still raises.)
#   'f32'    the f32-input MFMA, an exact fmaf chain
# NERO_GEMM=<mode> selects all passes, NERO_GEMM_FWD / _TAN / _BWD / _DW one pass.
_MODE_NAMES = {'f32': L.GEMM_F32, 'bf16x6': L.GEMM_BF16X6, 'f16x3': L.GEMM_F16X3}
_F16 = (L.GEMM_F16X3,)                        # the engine of the kind-3 packed images
_DEFAULT = {'fwd': 'f16x3', 'tan': 'f16x3', 'bwd': 'f16x3', 'dw': 'f16x3'}


def _resolve(mode, k):
    if mode not in _MODE_NAMES:
        raise ValueError(f"NERO_GEMM{'_' + k.upper()}: unknown arithmetic {mode!r} (one of {sorted(_MODE_NAMES)}; 'f16x3p' is not an arithmetic any more: "
                         "NERO_F16_PAIRED selects the workgroup organisation, DESIGN.md 9.3)")
    return _MODE_NAMES[mode]


GEMM_MODE = {k: _resolve(os.environ.get('NERO_GEMM_' + k.upper(), os.environ.get('NERO_GEMM', _DEFAULT[k])), k)
             for k in ('fwd', 'tan', 'bwd', 'dw')}


def set_gemm_mode(mode, passes=('fwd', 'tan', 'bwd', 'dw')):
    """select the dense-layer arithmetic ('f32' | 'bf16x6' | 'f16x3' | 'default') for the given passes; chains must be
    (re)packed afterwards.  'f16x3' on the weight-gradient pass means bf16x6."""
    for k in passes:
        GEMM_MODE[k] = _resolve(_DEFAULT[k] if mode == 'default' else mode, k)


def f16_paired(mask=-1):
    """workgroup organisation of the f16x3 chain passes (include/nero_hip.h::nero_f16_paired): bit 0 / 1 / 2 = forward / tangent / reverse on
    the two-workgroups-per-CU kernels for launches of more than 4 tiles per CU, bit 3 = for launches of every size; mask < 0 only queries.
    Returns the previous mask.  Results are bit-identical in every setting (tests/test_paired_engine.py)."""
    L.lib.nero_f16_paired.argtypes = [C.c_int]
    L.lib.nero_f16_paired.restype = C.c_int
    return int(L.lib.nero_f16_paired(int(mask)))


def f16_rowowner(mask=-1):
    """forward chains that save nothing on the row-owner kernel (include/nero_hip.h::nero_f16_rowowner, mlp_f16r.hip): bit 0 = launches of at
    least 128 rows per CU, bit 1 = every launch; mask < 0 only queries.  Returns the previous mask.  Bit-identical results
    (tests/test_rowowner_engine.py)."""
    L.lib.nero_f16_rowowner.argtypes = [C.c_int]
    L.lib.nero_f16_rowowner.restype = C.c_int
    return int(L.lib.nero_f16_rowowner(int(mask)))


# test hook (tests/test_parity_at_size.py, gate-teacher-forced gradient parity): when a list, every saving forward launch appends the
# ReLU sign masks its kernel wrote (nero_fwd_layer.relu_mask: one word per (row, 32-column tile)) with the chain's signature
MASK_CAPTURE = None


def decode_relu_masks(words, n_rows, n_out):
    """[rows_pad, 8] int32 sign words of one layer -> bool [n_rows, n_out]: gate[r, f] = (the layer's output f of row r is > 0).
    Bit 16 h + 4 g + j of word (r, t) is output 32 t + 8 g + 4 h + j (the accumulator layout of the chain kernels, mlp_f16x3.hip)."""
    import torch
    b = torch.arange(32, device=words.device)
    h, g, j = b >> 4, (b >> 2) & 3, b & 3
    perm = 8 * g + 4 * h + j                                        # bit -> output inside the tile
    bits = ((words[:n_rows].unsqueeze(-1) >> b) & 1).bool()        # [n, 8, 32] in bit order
    gate = torch.empty_like(bits)
    gate[:, :, perm] = bits
    return gate.reshape(n_rows, 256)[:, :n_out]


def _r8(x):
    return (x + 7) // 8 * 8


def _r16(x):
    return (x + 15) // 16 * 16


def _tiles(x):
    return (x + 31) // 32


def _rev_k(n_out):
    """contraction length of a layer's REVERSE GEMM (over its outputs) in the fp16 engines: wide layers are zero-padded to the full
    256 (the SDF's 217-wide layer 3: 224 -> 256, two all-zero k-steps) so that every reverse GEMM of the chain has 16 k-steps and
    the walk takes the fully unrolled kernel (bwd_f16_kernel<FIXED>, ~10 % faster)"""
    return 256 if 128 < n_out <= 256 else _r16(n_out)


def row_pad(n):
    return (n + 63) // 64 * 64


class Dense:
    """one Linear: W [n_out, k_total] (row-major, autograd tensor allowed), columns [main_c0, main_c0+k_main) multiply the
    resident activation tile, columns [aux_c0, aux_c0+k_aux) multiply the aux tile; `scale` is folded into the packed
    operand (SDF skip: 1/sqrt(2))."""

    def __init__(self, W, b, act, k_main, main_c0=0, k_aux=0, aux_c0=0, scale=1.0):
        self.W, self.b, self.act = W, b, act
        self.k_main, self.main_c0, self.k_aux, self.aux_c0, self.scale = k_main, main_c0, k_aux, aux_c0, scale
        self.n_out = W.shape[0]


class Head:
    """VALU head evaluated on the INPUT tile of its entry: W [n_head<=4, k<=256]."""

    def __init__(self, W, b):
        self.W, self.b = W, b
        self.n_head, self.k = W.shape


def run_pack_jobs(jobs):
    """launch a list of (PackJob, keep-alive) in as few nero_pack_batch calls as possible (<= MAX_PACK_JOBS jobs each)"""
    st = L.stream_ptr()
    for i0 in range(0, len(jobs), L.MAX_PACK_JOBS):
        chunk = jobs[i0:i0 + L.MAX_PACK_JOBS]
        arr = (L.PackJob * len(chunk))(*[j for j, _ in chunk])
        L.check(L.lib.nero_pack_batch(arr, len(chunk), st))


class Chain:
    def __init__(self, entries, k_init, k_aux=0, aux_wide=False, device='cuda'):
        """entries: list of (Dense|None, Head|None).  k_init / k_aux: padded widths (multiples of 4) of the init / aux
        matrices that are loaded into LDS."""
        assert len(entries) <= L.MAX_LAYERS
        self.entries, self.k_init, self.k_aux, self.aux_wide, self.device = entries, k_init, k_aux, aux_wide, device
        self.dense_idx = [i for i, (d, _) in enumerate(entries) if d is not None]
        self._packed = None

    # ------------------------------------------------------------------------------------------------------------
    def _sizes(self):
        """per entry: {image key: float count} for the current GEMM modes"""
        sizes = []
        for d, h in self.entries:
            e = {}
            if d is not None:
                nt = _tiles(d.n_out)
                if L.GEMM_F32 in (GEMM_MODE['fwd'], GEMM_MODE['tan']):
                    e['fm'] = (_r8(d.k_main) // 8) * nt * 256 if d.k_main else 0
                    e['fa'] = (_r8(d.k_aux) // 8) * nt * 256 if d.k_aux else 0
                if GEMM_MODE['bwd'] == L.GEMM_F32:
                    e['bm'] = (_r8(d.n_out) // 8) * _tiles(d.k_main) * 256 if d.k_main else 0
                    e['ba'] = (_r8(d.n_out) // 8) * _tiles(d.k_aux) * 256 if d.k_aux else 0
                e['bias'] = 32 * nt
                if L.GEMM_BF16X6 in (GEMM_MODE['fwd'], GEMM_MODE['tan']):   # three bf16 planes: 768 floats per (tile, 16-k step)
                    e['sfm'] = (_r16(d.k_main) // 16) * nt * 768 if d.k_main else 0
                    e['sfa'] = (_r16(d.k_aux) // 16) * nt * 768 if d.k_aux else 0
                if GEMM_MODE['fwd'] in _F16 or GEMM_MODE['tan'] in _F16:       # 64-float header + two fp16 planes: 512 floats per (tile, step)
                    e['hfm'] = 64 + (_r16(d.k_main) // 16) * nt * 512 if d.k_main else 0
                    e['hfa'] = 64 + (_r16(d.k_aux) // 16) * nt * 512 if d.k_aux else 0
                if GEMM_MODE['bwd'] in _F16:
                    e['hbm'] = 64 + (_rev_k(d.n_out) // 16) * _tiles(d.k_main) * 512 if d.k_main else 0
                    e['hba'] = 64 + (_rev_k(d.n_out) // 16) * _tiles(d.k_aux) * 512 if d.k_aux else 0
                if GEMM_MODE['bwd'] == L.GEMM_BF16X6:
                    e['sbm'] = (_r16(d.n_out) // 16) * _tiles(d.k_main) * 768 if d.k_main else 0
                    e['sba'] = (_r16(d.n_out) // 16) * _tiles(d.k_aux) * 768 if d.k_aux else 0
            if h is not None:
                e['hw'] = 4 * L.HID
                e['hb'] = 4
            sizes.append(e)
        return sizes

    def pack_floats(self):
        """size (floats) of this chain's packed operand images under the current GEMM modes"""
        return sum(sum(e.values()) for e in self._sizes())

    def pack(self, buf=None, run=True):
        """(re)build the packed operand images from the current effective weights (call once per optimisation step).
        buf: a ZERO-FILLED float32 buffer of pack_floats() elements to carve the images from (default: a fresh one).
        run=False: do not launch; return the list of (PackJob, keep-alive tensor) for a batched launch (run_pack_jobs)."""
        sizes = self._sizes()
        total = sum(sum(e.values()) for e in sizes)
        if buf is None:
            buf = torch.zeros(total, dtype=torch.float32, device=self.device)
        assert buf.numel() >= total
        off = 0
        packed = []
        jobs = []

        def job(kind, W, out, nrows, ld, col0, ncols, transpose, kpad, nt_count, scale=1.0):
            j = L.PackJob()
            j.W, j.out, j.kind, j.nrows, j.ld, j.col0, j.ncols = W.data_ptr(), out.data_ptr(), kind, nrows, ld, col0, ncols
            j.transpose, j.kpad, j.nt_count, j.scale = transpose, kpad, nt_count, scale
            jobs.append((j, W))

        for (d, h), e in zip(self.entries, sizes):
            p = {}
            for k, n in e.items():
                p[k] = buf[off:off + n] if n else None
                off += n
            if d is not None:
                W = d.W.detach()
                assert W.stride(1) == 1
                nt = _tiles(d.n_out)
                for key, c0, kc in (('fm', d.main_c0, d.k_main), ('fa', d.aux_c0, d.k_aux)):
                    if kc and key in p:
                        job(1, W, p[key], d.n_out, W.stride(0), c0, kc, 0, _r8(kc), nt, d.scale)
                for key, c0, kc in (('bm', d.main_c0, d.k_main), ('ba', d.aux_c0, d.k_aux)):
                    if kc and key in p:
                        job(1, W, p[key], d.n_out, W.stride(0), c0, kc, 1, _r8(d.n_out), _tiles(kc), d.scale)
                for key, c0, kc in (('sfm', d.main_c0, d.k_main), ('sfa', d.aux_c0, d.k_aux)):
                    if kc and key in p:
                        job(0, W, p[key], d.n_out, W.stride(0), c0, kc, 0, _r16(kc), nt, d.scale)
                for key, c0, kc in (('sbm', d.main_c0, d.k_main), ('sba', d.aux_c0, d.k_aux)):
                    if kc and key in p:
                        job(0, W, p[key], d.n_out, W.stride(0), c0, kc, 1, _r16(d.n_out), _tiles(kc), d.scale)
                for key, c0, kc in (('hfm', d.main_c0, d.k_main), ('hfa', d.aux_c0, d.k_aux)):
                    if kc and key in p:
                        job(3, W, p[key], d.n_out, W.stride(0), c0, kc, 0, _r16(kc), nt, d.scale)
                for key, c0, kc in (('hbm', d.main_c0, d.k_main), ('hba', d.aux_c0, d.k_aux)):
                    if kc and key in p:
                        job(3, W, p[key], d.n_out, W.stride(0), c0, kc, 1, _rev_k(d.n_out), _tiles(kc), d.scale)
                if d.b is not None:
                    b = d.b.detach()
                    job(2, b, p['bias'], 1, d.n_out, 0, d.n_out, 0, 32 * nt, 0)
            if h is not None:
                Wh = h.W.detach()
                assert Wh.stride(1) == 1
                job(2, Wh, p['hw'], h.n_head, Wh.stride(0), 0, h.k, 0, L.HID, 0)
                if h.b is not None:
                    job(2, h.b.detach(), p['hb'], 1, h.n_head, 0, h.n_head, 0, 4, 0)
            packed.append(p)
        self._packed, self._buf = packed, buf
        if not run:
            return jobs
        run_pack_jobs(jobs)
        return self

    # ------------------------------------------------------------------------------------------------------------
    def forward(self, init, aux, n_rows, save=True):
        """-> dict(saves=[per dense entry: [rows_pad,256] or None], heads={entry: [rows_pad,4]})"""
        assert self._packed is not None
        rp = row_pad(n_rows)
        ch = L.FwdChain()
        ch.init, ch.ld_init, ch.k_init = L.ptr(init), (init.stride(0) if init is not None else 0), self.k_init
        ch.aux, ch.ld_aux, ch.k_aux = L.ptr(aux), (aux.stride(0) if aux is not None else 0), self.k_aux
        ch.n_layers, ch.aux_wide = len(self.entries), int(self.aux_wide)
        split = GEMM_MODE['fwd'] != L.GEMM_F32
        fkeys = {L.GEMM_F32: ('fm', 'fa'), L.GEMM_BF16X6: ('sfm', 'sfa'), L.GEMM_F16X3: ('hfm', 'hfa')}[GEMM_MODE['fwd']]
        ch.gemm_mode = GEMM_MODE['fwd']
        ch.macs_per_row = float(sum(d.n_out * (d.k_main + d.k_aux) for d, _ in self.entries if d is not None))
        if init is not None:
            assert init.shape[0] >= rp and init.shape[1] >= self.k_init
        if aux is not None:
            assert aux.shape[0] >= rp and aux.shape[1] >= self.k_aux
        nd = len(self.dense_idx)
        last_dense = self.dense_idx[-1] if nd else -1
        saves, heads = [None] * len(self.entries), {}
        n_save = sum(1 for i in self.dense_idx if save or i == last_dense)
        sbuf = torch.empty((n_save, rp, L.HID), dtype=torch.float32, device=self.device)
        si = 0
        # ReLU sign masks (32 B per row and layer) for the reverse pass of the fp16 tile engine: it then skips the 1 KiB/row read of
        # the saved activation (the weight-gradient GEMM still reads it)
        masks = [None] * len(self.entries)
        relu_idx = [i for i in self.dense_idx if save and self.entries[i][0].act == L.ACT_RELU] if GEMM_MODE['fwd'] in _F16 else []
        mbuf = torch.empty((len(relu_idx), rp, 8), dtype=torch.int32, device=self.device) if relu_idx else None
        for k, i in enumerate(relu_idx):
            masks[i] = mbuf[k]
        for i, ((d, h), p) in enumerate(zip(self.entries, self._packed)):
            fl = ch.layer[i]
            if h is not None:
                ho = torch.empty((rp, 4), dtype=torch.float32, device=self.device)
                heads[i] = ho
                fl.head_w, fl.head_b, fl.head_out = p['hw'].data_ptr(), p['hb'].data_ptr(), ho.data_ptr()
                fl.n_head, fl.head_k = h.n_head, (h.k + 3) // 4 * 4
            if d is not None:
                rk = _r16 if split else _r8
                fl.w_main = L.ptr(p.get(fkeys[0]))
                fl.w_aux = L.ptr(p.get(fkeys[1]))
                fl.bias = p['bias'].data_ptr()
                fl.k_main, fl.k_aux = (rk(d.k_main) if d.k_main else 0), (rk(d.k_aux) if d.k_aux else 0)
                fl.n_tiles, fl.act = _tiles(d.n_out), d.act
                if save or i == last_dense:
                    saves[i] = sbuf[si]
                    si += 1
                    fl.save = saves[i].data_ptr()
                if masks[i] is not None:
                    fl.relu_mask = masks[i].data_ptr()
        L.check(L.lib.nero_mlp_forward(C.byref(ch), n_rows, L.stream_ptr()))
        if MASK_CAPTURE is not None and relu_idx:
            MASK_CAPTURE.append({'k_init': self.k_init, 'k_aux': self.k_aux, 'aux_wide': bool(self.aux_wide), 'n_rows': n_rows,
                                 'n_out': [d.n_out if d is not None else 0 for d, _ in self.entries], 'masks': masks})
        return {'saves': saves, 'heads': heads, 'masks': masks, '_keep': (init, aux)}

    # ------------------------------------------------------------------------------------------------------------
    def backward(self, fwd, n_rows, dy=None, head_dys=None, need_dinit=False, need_daux=False, injs=None,
                 dinit_out=None, accumulate_dinit=False, skip_last_dense=False):
        """reverse pass.  dy: [rows_pad, >=n_out_last] gradient w.r.t. the last dense output (or w.r.t. the last
        pseudo-entry's input tile).  head_dys: {entry: [rows_pad,4]}.  injs: {dense entry: [rows_pad,256]} added to the
        delta of that entry's OUTPUT -- or {dense entry: (gbar, adot)}: the fp16 engine then forms the sigma'' injection of a
        softplus entry itself (nero_bwd_layer.inj_adot).  -> dict(deltas={entry: [rows_pad,256]}, d_init, d_aux)"""
        head_dys = head_dys or {}
        injs = injs or {}
        rp = row_pad(n_rows)
        saves = fwd['saves']
        ch = L.BwdChain()
        ch.n_layers, ch.aux_wide = len(self.entries), 0
        split = GEMM_MODE['bwd'] != L.GEMM_F32
        bkeys = {L.GEMM_F32: ('bm', 'ba'), L.GEMM_BF16X6: ('sbm', 'sba'), L.GEMM_F16X3: ('hbm', 'hba')}[GEMM_MODE['bwd']]
        ch.gemm_mode = GEMM_MODE['bwd']
        rk = _r16 if split else _r8
        last = len(self.entries) - 1
        if dy is not None:
            ch.dy, ch.ld_dy = dy.data_ptr(), dy.stride(0)
            d_last = self.entries[last][0]
            ch.k_dy = _r8(d_last.n_out) if d_last is not None else L.HID
            assert dy.shape[1] >= ch.k_dy
        d_init = d_aux = None
        if need_dinit:
            d_init = dinit_out if dinit_out is not None else torch.empty((rp, self.k_init), dtype=torch.float32, device=self.device)
            ch.d_init, ch.ld_dinit, ch.accumulate_dinit = d_init.data_ptr(), d_init.stride(0), int(accumulate_dinit)
        if need_daux:
            d_aux = torch.zeros((rp, self.k_aux), dtype=torch.float32, device=self.device)
            ch.d_aux, ch.ld_daux = d_aux.data_ptr(), d_aux.stride(0)
        deltas = {}
        n_delta = len(self.dense_idx)
        dbuf = torch.empty((n_delta, rp, L.HID), dtype=torch.float32, device=self.device)
        prev_dense = {}
        pd = None
        for i, (d, h) in enumerate(self.entries):
            prev_dense[i] = pd
            if d is not None:
                pd = i
        for k, i in enumerate(self.dense_idx):
            deltas[i] = dbuf[k]
        for i, ((d, h), p) in enumerate(zip(self.entries, self._packed)):
            bl = ch.layer[i]
            j = prev_dense[i]                      # dense entry that produced this entry's input tile
            if d is not None and not (skip_last_dense and i == last):
                bl.w_main_t = L.ptr(p.get(bkeys[0]))
                bl.w_aux_t = L.ptr(p.get(bkeys[1])) if need_daux else None
                bl.n_out = _rev_k(d.n_out) if GEMM_MODE['bwd'] in _F16 else rk(d.n_out)
                bl.k_main_tiles = _tiles(d.k_main) if d.k_main else 0
                bl.k_aux_tiles = _tiles(d.k_aux) if d.k_aux else 0
            else:
                bl.n_out = 0
                bl.k_main_tiles = _tiles(self.entries[j][0].n_out)
            if h is not None and i in head_dys:
                bl.head_w, bl.head_dy, bl.n_head = p['hw'].data_ptr(), head_dys[i].data_ptr(), h.n_head
            if j is not None:
                bl.a_prev = saves[j].data_ptr()
                bl.act_prev = self.entries[j][0].act
                mk = fwd.get('masks')
                if mk is not None and mk[j] is not None and GEMM_MODE['bwd'] in _F16:
                    bl.mask_prev = mk[j].data_ptr()
                bl.delta_prev = deltas[j].data_ptr()
                if j in injs:
                    if isinstance(injs[j], tuple):
                        assert GEMM_MODE['bwd'] in _F16 and self.entries[j][0].act == L.ACT_SOFTPLUS100
                        bl.inj, bl.inj_adot = injs[j][0].data_ptr(), injs[j][1].data_ptr()
                    else:
                        bl.inj = injs[j].data_ptr()
        # the delta of the last dense entry is dy itself when that entry is the chain's last entry
        if self.entries[last][0] is not None and not skip_last_dense:
            assert dy is not None and self.entries[last][0].act == L.ACT_NONE
            deltas[last] = dy
        macs = 0.0
        for i, (d, h) in enumerate(self.entries):
            if d is None or (skip_last_dense and i == last):
                continue
            first = prev_dense[i] is None
            if first and not need_dinit:
                macs += d.n_out * d.k_aux if (need_daux and d.k_aux) else 0
            else:
                macs += d.n_out * (d.k_main + (d.k_aux if need_daux else 0))
        ch.macs_per_row = float(macs)
        L.check(L.lib.nero_mlp_backward(C.byref(ch), n_rows, L.stream_ptr()))
        return {'deltas': deltas, 'd_init': d_init, 'd_aux': d_aux}

    # ------------------------------------------------------------------------------------------------------------
    def weight_grads(self, fwd, bwd, n_rows, init, aux, head_dys=None, workspace=None, second=None, head_extra=None, outs=None):
        """-> list per entry of dict(dW, db, dWh, dbh) (torch tensors shaped like the effective weights).
        second: optional {dense entry: (D1 [rows,256], B1_main [rows,256] or init-like, B1_aux)} extra operand pair
        accumulated into the same dW (SDF double-backward).
        outs: optional {dense entry: (dW_out [n_out, k_total], db_out [n_out])} -- caller-owned destinations (e.g. views of a flat
        gradient bucket) the GEMM writes straight into instead of fresh tensors."""
        head_dys = head_dys or {}
        second = second or {}
        head_extra = head_extra or {}
        if workspace is None:
            workspace = torch.empty(L.lib.nero_dw_workspace_floats(max(n_rows, 1)), dtype=torch.float32, device=self.device)
        st = L.stream_ptr()
        out = []
        prev = None
        jobs = []                      # every dense layer's job(s): issued together at the end (nero_dw_gemm_batch)
        for i, (d, h) in enumerate(self.entries):
            g = {}
            if h is not None and i in head_dys:
                a_in = fwd['saves'][prev]
                g['dWh'] = torch.empty((4, L.HID), dtype=torch.float32, device=self.device)
                g['dbh'] = torch.empty(4, dtype=torch.float32, device=self.device)
                L.check(L.lib.nero_head_dw(C.c_void_p(head_dys[i].data_ptr()), C.c_void_p(a_in.data_ptr()),
                                           C.c_void_p(L.ptr(head_extra.get(i))), h.n_head, n_rows,
                                           C.c_void_p(g['dWh'].data_ptr()), C.c_void_p(g['dbh'].data_ptr()),
                                           C.c_void_p(workspace.data_ptr()), 0, st))
                g['dWh'] = g['dWh'][:h.n_head, :h.k]
                g['dbh'] = g['dbh'][:h.n_head]
            if d is not None:
                delta = bwd['deltas'][i]
                if outs is not None and i in outs:
                    dW, db = outs[i]
                    assert dW.shape == d.W.shape and dW.stride(1) == 1 and db.numel() == d.n_out
                else:
                    dW = torch.empty_like(d.W.detach())
                    db = torch.empty(d.n_out, dtype=torch.float32, device=self.device)
                main_in = init if prev is None else fwd['saves'][prev]
                sec = second.get(i)
                parts = []
                if d.k_main:
                    parts.append((main_in, d.k_main, d.main_c0, 0))
                if d.k_aux:
                    parts.append((aux, d.k_aux, d.aux_c0, 1))
                for pi, (Bm, kc, c0, which) in enumerate(parts):
                    job = L.DwJob()
                    job.d0, job.ldd0, job.b0, job.ldb0 = delta.data_ptr(), delta.stride(0), Bm.data_ptr(), Bm.stride(0)
                    if sec is not None:
                        D1, B1 = sec[0], sec[1 + which]
                        job.d1, job.ldd1, job.b1, job.ldb1 = D1.data_ptr(), D1.stride(0), B1.data_ptr(), B1.stride(0)
                    job.n_out, job.k_cols = d.n_out, kc
                    job.dW, job.ldw, job.col0 = dW.data_ptr(), dW.stride(0), c0
                    job.db = db.data_ptr() if pi == 0 else None
                    job.scale, job.accumulate = d.scale, 0
                    job.gemm_mode = GEMM_MODE['dw']
                    jobs.append(job)
                g['dW'], g['db'] = dW, db
                prev = i
            out.append(g)
        if jobs:
            arr = (L.DwJob * len(jobs))(*jobs)
            L.check(L.lib.nero_dw_gemm_batch(arr, len(jobs), n_rows, C.c_void_p(workspace.data_ptr()), st))
        return out
