"""ctypes binding of libnero_hip.so (include/nero_hip.h).  The library is REQUIRED: there is no PyTorch/CPU fallback for
the product path -- if it is missing or fails to load, importing this module raises."""
import ctypes as C
import os

import torch  # noqa: F401  -- must come first: libnero_hip.so has to bind to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('NERO_HIP_LIB') or os.path.join(_HERE, 'libnero_hip.so')     # (override: kernel-variant experiments)

MAX_LAYERS = 10
HID = 256
ACT_NONE, ACT_RELU, ACT_SOFTPLUS100 = 0, 1, 2
GEMM_F32, GEMM_BF16X6, GEMM_F16X3 = 0, 1, 2

_fp = C.c_void_p   # device pointers travel as integers


class FwdLayer(C.Structure):
    _fields_ = [('w_main', _fp), ('w_aux', _fp), ('bias', _fp), ('save', _fp), ('head_w', _fp), ('head_b', _fp),
                ('head_out', _fp), ('k_main', C.c_int), ('k_aux', C.c_int), ('n_tiles', C.c_int), ('n_head', C.c_int),
                ('act', C.c_int), ('head_k', C.c_int), ('relu_mask', _fp)]


class FwdChain(C.Structure):
    _fields_ = [('init', _fp), ('aux', _fp), ('ld_init', C.c_int), ('k_init', C.c_int), ('ld_aux', C.c_int),
                ('k_aux', C.c_int), ('n_layers', C.c_int), ('aux_wide', C.c_int), ('gemm_mode', C.c_int), ('pad_', C.c_int),
                ('macs_per_row', C.c_double), ('layer', FwdLayer * MAX_LAYERS)]


class TanLayer(C.Structure):
    _fields_ = [('w_main', _fp), ('w_aux', _fp), ('a_saved', _fp), ('gbar', _fp), ('adot', _fp), ('inj', _fp),
                ('k_main', C.c_int), ('k_aux', C.c_int), ('n_tiles', C.c_int), ('pad_', C.c_int)]


class TanChain(C.Structure):
    _fields_ = [('init', _fp), ('aux', _fp), ('ld_init', C.c_int), ('k_init', C.c_int), ('ld_aux', C.c_int),
                ('k_aux', C.c_int), ('n_layers', C.c_int), ('aux_wide', C.c_int), ('gemm_mode', C.c_int), ('pad_', C.c_int),
                ('macs_per_row', C.c_double), ('layer', TanLayer * MAX_LAYERS)]


class BwdLayer(C.Structure):
    _fields_ = [('w_main_t', _fp), ('w_aux_t', _fp), ('a_prev', _fp), ('inj', _fp), ('delta_prev', _fp),
                ('head_w', _fp), ('head_dy', _fp), ('n_out', C.c_int), ('k_main_tiles', C.c_int),
                ('k_aux_tiles', C.c_int), ('n_head', C.c_int), ('act_prev', C.c_int), ('pad_', C.c_int), ('mask_prev', _fp), ('inj_adot', _fp)]


class BwdChain(C.Structure):
    _fields_ = [('dy', _fp), ('ld_dy', C.c_int), ('k_dy', C.c_int), ('d_init', _fp), ('d_aux', _fp),
                ('ld_dinit', C.c_int), ('ld_daux', C.c_int), ('accumulate_dinit', C.c_int), ('n_layers', C.c_int),
                ('aux_wide', C.c_int), ('gemm_mode', C.c_int), ('macs_per_row', C.c_double), ('layer', BwdLayer * MAX_LAYERS)]


class PackJob(C.Structure):
    _fields_ = [('W', _fp), ('out', _fp), ('kind', C.c_int), ('nrows', C.c_int), ('ld', C.c_int), ('col0', C.c_int),
                ('ncols', C.c_int), ('transpose', C.c_int), ('kpad', C.c_int), ('nt_count', C.c_int), ('scale', C.c_float),
                ('pad_', C.c_int)]


MAX_PACK_JOBS = 64


class DwJob(C.Structure):
    _fields_ = [('d0', _fp), ('b0', _fp), ('d1', _fp), ('b1', _fp), ('ldd0', C.c_int), ('ldb0', C.c_int),
                ('ldd1', C.c_int), ('ldb1', C.c_int), ('n_out', C.c_int), ('k_cols', C.c_int), ('dW', _fp),
                ('ldw', C.c_int), ('col0', C.c_int), ('db', _fp), ('scale', C.c_float), ('accumulate', C.c_int),
                ('gemm_mode', C.c_int), ('pad_', C.c_int)]


class WnJob(C.Structure):
    _fields_ = [('v', _fp), ('g', _fp), ('w_eff', _fp), ('inv_norm', _fp), ('v_rw', _fp), ('g_rw', _fp), ('dW', _fp), ('m_v', _fp),
                ('v_v', _fp), ('m_g', _fp), ('v_g', _fp), ('rows', C.c_int), ('cols', C.c_int)]


class WnGradJob(C.Structure):
    _fields_ = [('v', _fp), ('g', _fp), ('inv_norm', _fp), ('dW', _fp), ('dv', _fp), ('dg', _fp), ('rows', C.c_int), ('cols', C.c_int)]


class AdamJob(C.Structure):
    _fields_ = [('p', _fp), ('grad', _fp), ('m', _fp), ('v', _fp), ('n', C.c_int), ('step', C.c_int)]


MAX_WN_JOBS, MAX_ADAM_JOBS = 40, 96


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                          f'(hipcc --offload-arch=gfx950).  nero_amd has no non-HIP fallback.')
    lib = C.CDLL(LIB_PATH)
    lib.nero_last_error.restype = C.c_char_p
    return lib


lib = _load()


class NeroHipError(RuntimeError):
    pass


class NeroOutOfMemory(NeroHipError, MemoryError):
    """a step workspace larger than what the device can hold (NERO_ERR_NOMEM, nero_check_device_memory)"""


def check_workspace_fits(need_bytes, device, held_bytes=0, what='step workspace'):
    """raise NeroOutOfMemory -- with the byte counts -- BEFORE torch is asked for a workspace that cannot fit: free device memory + what torch's
    caching allocator holds without using + the caller's previous workspace (released for the new one) must cover it"""
    import ctypes as C
    import torch
    reusable = int(held_bytes)
    try:
        reusable += max(0, torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device))
    except Exception:                                  # noqa: BLE001 (no CUDA context yet)
        pass
    lib.nero_check_device_memory.argtypes = [C.c_size_t, C.c_size_t, C.c_char_p]
    with torch.cuda.device(device):
        check(lib.nero_check_device_memory(int(need_bytes), reusable, what.encode()))


def check(rc):
    if rc != 0:
        msg = lib.nero_last_error().decode()
        if rc == -3:
            raise NotImplementedError(msg)
        if rc == -4:                                   # NERO_ERR_NOMEM: a workspace that cannot fit, reported BEFORE the allocation is tried
            raise NeroOutOfMemory(msg)
        raise NeroHipError(f'libnero_hip error {rc}: {msg}')


def ptr(t):
    """device pointer of a torch tensor (None -> NULL)"""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
