"""SDF field on the HIP engine: value + feature + normal (first-order input gradient) in the forward direction and the
second-order weight gradient in the reverse direction:

  forward :  z_l = W_l abar_{l-1} + b_l ; a_l = softplus100(z_l)                      (chain forward, saves a_l)
  normal  :  gbar_{l-1} = (W_l^T gbar_l) * s_{l-1},  gbar seeded by the sdf row of W_8 (chain reverse, saves gbar_l)
             n = J_e^T ebar
  reverse :  given dL/dsdf, dL/dfeat, dL/dn:
             tangent   adot_l = s_l * (W_l adot_{l-1}),  adot_{-1} = J_e dL/dn        (chain tangent)
             zhat_{l-1} = (W_l^T zhat_l) * s_{l-1} + gbar_{l-1} * beta (1-s_{l-1}) * zdot_{l-1}
             dW_l = zhat_l abar_{l-1}^T + gbar_l adotbar_{l-1}^T ,  db_l = sum zhat_l
Replaces SDFNetwork.forward/.gradient (network/field.py:130-167) and autograd's double backward through them.
"""
import ctypes as C
import math

import torch

from . import _lib as L
from .chain import Chain, Dense, Head, row_pad, _r8, _r16, _tiles, GEMM_MODE

N_FREQ, D_PE, LD_PE = 6, 39, 40                        # the shipped YAMLs: sdf_freq 6 (network/renderer.py:75)
MAX_SDF_FREQ = 6                                        # the chain kernels' narrow aux tile and nero_pe_vjp / _jvp hold 40 columns: 3 + 6 f <= 40


def sdf_shape(eff):
    """(n_lin, d_pe, n_freq, ld_pe, skip) of an SDF network given as its list of effective (W, b): sdf_n_layers + 1 linear layers, PE-f
    input of 3 + 6 f columns (rows padded to a multiple of 8 floats), the input re-injected in front of layer sdf_n_layers // 2
    (network/renderer.py:118-124, network/field.py:75-101)"""
    n_lin, d_pe = len(eff), int(eff[0][0].shape[1])
    assert (d_pe - 3) % 6 == 0 and d_pe > 3, d_pe
    return n_lin, d_pe, (d_pe - 3) // 6, (d_pe + 7) // 8 * 8, (n_lin - 1) // 2


def sdf_entries(eff):
    """eff: list of sdf_n_layers + 1 (W, b) effective weights (9 in every shipped YAML)."""
    n_lin, d_pe, _, _, skip = sdf_shape(eff)
    e = []
    for l, (W, b) in enumerate(eff):
        if l == 0:
            e.append((Dense(W, b, L.ACT_SOFTPLUS100, d_pe), None))
        elif l == skip:
            e.append((Dense(W, b, L.ACT_SOFTPLUS100, 256 - d_pe, 0, d_pe, 256 - d_pe, 1.0 / math.sqrt(2)), None))
        elif l == n_lin - 1:
            e.append((Dense(W[1:], b[1:], L.ACT_NONE, 256), Head(W[0:1], b[0:1])))
        else:
            e.append((Dense(W, b, L.ACT_SOFTPLUS100, 256), None))
    return e


def encode_pe(x, n, dim, n_freq, ld):
    out = torch.empty((row_pad(n), ld), dtype=torch.float32, device=x.device)
    L.check(L.lib.nero_encode_pe(C.c_void_p(x.data_ptr()), x.stride(0), dim, n_freq, n, C.c_void_p(out.data_ptr()), ld, L.stream_ptr()))
    return out


class SDFField:
    def __init__(self, eff, device='cuda'):
        self.device = device
        self.n_lin, self.d_pe, self.n_freq, self.ld_pe, self.skip = sdf_shape(eff)
        self.last = last = self.n_lin - 1
        # the fused point encoders (nero_ray_points_pe, nero_gather_inner) and the C step driver are written for PE-6 rows of 40 floats;
        # any other sdf_freq goes through nero_encode_pe (pe_of_rays / RenderCore)
        self.default_pe = (self.n_freq == N_FREQ)
        self.full = Chain(sdf_entries(eff), k_init=self.ld_pe, k_aux=self.ld_pe, device=device)
        self.value_only = Chain(self.full.entries[:last] + [(None, self.full.entries[last][1])], k_init=self.ld_pe, k_aux=self.ld_pe, device=device)
        self._ones = None

    def pack(self, buf=None, run=True):
        jobs = self.full.pack(buf, run)
        last = self.last
        self.value_only._packed = self.full._packed[:last] + [{k: v for k, v in self.full._packed[last].items() if k in ('hw', 'hb')}]
        return self if run else jobs

    def pe_of_rays(self, o, d, z, col0, ncols):
        """PE rows [row_pad(R * ncols), ld_pe] of the points o + z[:, col0 : col0 + ncols] d (ray-major).  PE-6: the fused kernel."""
        R = o.shape[0]
        if self.default_pe:
            pe = torch.empty((row_pad(R * ncols), LD_PE), dtype=torch.float32, device=o.device)
            L.check(L.lib.nero_ray_points_pe(C.c_void_p(o.data_ptr()), C.c_void_p(d.data_ptr()), C.c_void_p(z.data_ptr()), z.stride(0), col0, ncols, R,
                                             C.c_void_p(pe.data_ptr()), L.stream_ptr()))
            return pe
        pts = (o[:, None, :] + z[:, col0:col0 + ncols, None] * d[:, None, :]).reshape(R * ncols, 3).contiguous()
        return encode_pe(pts, R * ncols, 3, self.n_freq, self.ld_pe)

    def pack_floats(self):
        return self.full.pack_floats()

    # -- no-grad value evaluation (sampler, occ-loss march, mesh extraction) --------------------------------------
    def sdf_from_pe(self, pe, n):
        """pe: [rows_pad, 40] -> [rows_pad, 4] tensor whose column 0 is the sdf"""
        return self.value_only.forward(pe, pe, n, save=False)['heads'][self.last]

    def sdf(self, x):
        n = x.shape[0]
        return self.sdf_from_pe(encode_pe(x.contiguous(), n, 3, self.n_freq, self.ld_pe), n)[:n, 0:1]

    # -- value + feature + normal ----------------------------------------------------------------------------------
    def forward_normal(self, x, n, pe=None):
        """x [n,>=3] contiguous rows (ld = x.stride(0)).  -> ctx dict with sdf [rows_pad,4](col 0), feat [rows_pad,256],
        normal [n,3]"""
        rp = row_pad(n)
        last = self.last
        if pe is None:
            pe = encode_pe(x, n, 3, self.n_freq, self.ld_pe)
        fwd = self.full.forward(pe, pe, n, save=True)
        if self._ones is None or self._ones.shape[0] < rp:
            self._ones = torch.zeros((rp, 4), dtype=torch.float32, device=self.device)
            self._ones[:, 0] = 1.0
        nb = self.full.backward(fwd, n, dy=None, head_dys={last: self._ones}, need_dinit=True, need_daux=True, skip_last_dense=True)
        normal = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        L.check(L.lib.nero_pe_vjp(C.c_void_p(x.data_ptr()), x.stride(0), C.c_void_p(nb['d_init'].data_ptr()), nb['d_init'].stride(0),
                                  C.c_void_p(nb['d_aux'].data_ptr()), nb['d_aux'].stride(0), self.n_freq, n,
                                  C.c_void_p(normal.data_ptr()), 3, L.stream_ptr()))
        return {'x': x, 'n': n, 'pe': pe, 'fwd': fwd, 'gbar': nb['deltas'], 'sdf4': fwd['heads'][last], 'feat': fwd['saves'][last],
                'normal': normal}

    # -- reverse of (sdf, feat, normal) w.r.t. the weights -----------------------------------------------------------
    def backward(self, ctx, d_sdf4, d_feat, d_normal, workspace=None, outs=None):
        """d_sdf4 [rows_pad,4] (col 0 used), d_feat [rows_pad,256], d_normal [n,3] or None.
        -> list of n_lin (dW [n_out,k], db [n_out]) for lin0..lin<last> (the last with all 257 rows)."""
        n, x, pe, fwd, gbar = ctx['n'], ctx['x'], ctx['pe'], ctx['fwd'], ctx['gbar']
        rp = row_pad(n)
        ch = self.full
        last, LD_PE, N_FREQ = self.last, self.ld_pe, self.n_freq
        injs, second, head_extra = {}, {}, {}
        if d_normal is not None:
            ehat = torch.empty((rp, LD_PE), dtype=torch.float32, device=self.device)
            L.check(L.lib.nero_pe_jvp(C.c_void_p(x.data_ptr()), x.stride(0), C.c_void_p(d_normal.data_ptr()), d_normal.stride(0),
                                      N_FREQ, n, C.c_void_p(ehat.data_ptr()), LD_PE, L.stream_ptr()))
            tc = L.TanChain()
            tc.init, tc.ld_init, tc.k_init = ehat.data_ptr(), LD_PE, LD_PE
            tc.aux, tc.ld_aux, tc.k_aux = ehat.data_ptr(), LD_PE, LD_PE
            tc.n_layers, tc.aux_wide = last, 0
            tsplit = GEMM_MODE['tan'] != L.GEMM_F32
            tkeys = {L.GEMM_F32: ('fm', 'fa'), L.GEMM_BF16X6: ('sfm', 'sfa'), L.GEMM_F16X3: ('hfm', 'hfa')}[GEMM_MODE['tan']]
            tc.gemm_mode = GEMM_MODE['tan']
            rk = _r16 if tsplit else _r8
            tc.macs_per_row = float(sum(ch.entries[l][0].n_out * (ch.entries[l][0].k_main + ch.entries[l][0].k_aux) for l in range(last)))
            # fp16 engines: the reverse kernel forms the injections from (gbar, adot) itself -- the tangent pass then reads no gbar and
            # writes no inj (4 -> 2 KB per row and layer; the reverse pass reads 1 KB more)
            fused_inj = GEMM_MODE['tan'] == L.GEMM_F16X3 and GEMM_MODE['bwd'] == L.GEMM_F16X3
            tbuf = torch.empty((1 if fused_inj else 2, last, rp, L.HID), dtype=torch.float32, device=self.device)
            for l in range(last):
                d, p = ch.entries[l][0], ch._packed[l]
                tl = tc.layer[l]
                tl.w_main, tl.w_aux = L.ptr(p.get(tkeys[0])), L.ptr(p.get(tkeys[1]))
                tl.a_saved, tl.gbar = fwd['saves'][l].data_ptr(), (None if fused_inj else gbar[l].data_ptr())
                tl.adot, tl.inj = tbuf[0, l].data_ptr(), (None if fused_inj else tbuf[1, l].data_ptr())
                tl.k_main, tl.k_aux, tl.n_tiles = rk(d.k_main), (rk(d.k_aux) if d.k_aux else 0), _tiles(d.n_out)
                injs[l] = (gbar[l], tbuf[0, l]) if fused_inj else tbuf[1, l]
            L.check(L.lib.nero_mlp_tangent(C.byref(tc), n, L.stream_ptr()))
            for l in range(last):
                second[l] = (gbar[l], ehat if l == 0 else tbuf[0, l - 1], ehat)
            head_extra[last] = tbuf[0, last - 1]
        bwd = ch.backward(fwd, n, dy=d_feat, head_dys={last: d_sdf4}, injs=injs)
        gr = ch.weight_grads(fwd, bwd, n, pe, pe, head_dys={last: d_sdf4}, workspace=workspace, second=second, head_extra=head_extra, outs=outs)
        out = []
        for l in range(self.n_lin):
            if l == last:
                out.append((torch.cat([gr[last]['dWh'], gr[last]['dW']], 0), torch.cat([gr[last]['dbh'], gr[last]['db']], 0)))
            else:
                out.append((gr[l]['dW'], gr[l]['db']))
        return out
