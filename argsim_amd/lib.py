"""ctypes binding of libargsim_vae.so (include/argsim_vae.h).

The library is the product; there is NO CPU fallback.  If the shared object is missing or no
HIP device is present, the functions here raise."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libargsim_vae.so')
CSRC = os.path.join(_HERE, 'csrc')


def source_digest():
    """sha1 over the kernel and host sources the library is built from (csrc/*.hip, *.cpp, *.h and the C-ABI header): names a
    build.  The committed rocprofv3 summaries under profiles/ record it (scripts/summarize_profile.py) and bench.py only quotes
    their counters for the SAME sources -- the GPU box has no .git to ask for a commit."""
    import glob
    import hashlib
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.cpp')) + glob.glob(os.path.join(CSRC, '*.h')))
    files.append(os.path.join(os.path.dirname(_HERE), 'include', 'argsim_vae.h'))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


class AvaeConfig(C.Structure):
    _fields_ = [('dim_tgt', C.c_int32), ('dim_emb', C.c_int32), ('dim_rep', C.c_int32), ('rnn_layers', C.c_int32),
                ('accelerate', C.c_float), ('learn_rate', C.c_float), ('bos', C.c_int32), ('eos', C.c_int32),
                ('max_batch', C.c_int32), ('max_len', C.c_int32), ('kl_beta', C.c_float), ('free_bits', C.c_float),
                ('compute_dtype', C.c_int32)]


GRAD_HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int64, C.c_int64)

_P = C.c_void_p
# name -> (restype, argtypes); exactly the declarations of include/argsim_vae.h
SIGNATURES = {
    'avae_create': (C.c_int, [C.POINTER(AvaeConfig), C.c_int, C.POINTER(_P)]),
    'avae_destroy': (None, [_P]),
    'avae_last_error': (C.c_char_p, [_P]),
    'avae_set_stream': (C.c_int, [_P, _P]),
    'avae_get_dims': (C.c_int, [_P] + [C.POINTER(C.c_int32)] * 4),
    'avae_state_numel': (C.c_int64, [_P]),
    'avae_bind_state': (C.c_int, [_P, _P, _P, _P, _P]),
    'avae_param_count': (C.c_int, [_P]),
    'avae_param_name': (C.c_char_p, [_P, C.c_int]),
    'avae_param_info': (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    'avae_get_tensor': (C.c_int, [_P, C.c_char_p, C.c_int, _P]),
    'avae_set_tensor': (C.c_int, [_P, C.c_char_p, C.c_int, _P]),
    'avae_get_step': (C.c_int, [_P, C.POINTER(C.c_int64)]),
    'avae_set_step': (C.c_int, [_P, C.c_int64]),
    'avae_get_schedule': (C.c_int, [_P, C.POINTER(C.c_float)]),
    'avae_forward_backward': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, _P, _P, C.c_float, C.c_float]),
    'avae_adam_step': (C.c_int, [_P]),
    'avae_train_step': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, _P, _P]),
    'avae_get_losses': (C.c_int, [_P, C.POINTER(C.c_float)]),
    'avae_set_grad_hook': (C.c_int, [_P, GRAD_HOOK, _P]),
    'avae_eval': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.POINTER(C.c_int32)]),
    'avae_encode': (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P]),
    'avae_decode_init': (C.c_int, [_P, _P, C.c_int32, _P]),
    'avae_decode_step': (C.c_int, [_P, _P, _P, C.c_int32, _P, _P]),
    'avae_decode_greedy': (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, C.POINTER(C.c_int32)]),
    # knobs used by tests / bench (not part of the reference-facing surface)
    'avae_set_option': (C.c_int, [_P, C.c_char_p, C.c_int]),
    'avae_debug_gemm': (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P] + [C.c_int] * 6 + [C.c_float, C.c_int, C.c_int]),
    'avae_debug_gemm_dyn': (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    'avae_debug_gemm_c16': (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, _P]),
    'avae_debug_gemm_tn16': (C.c_int, [_P, _P, _P, _P] + [C.c_int] * 6 + [C.c_float]),
    'avae_timing_collect': (C.c_int, [_P, C.POINTER(C.c_double)]),
    'avae_debug_timing': (C.c_int, [_P, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]),
    'avae_debug_train_ce': (C.c_int, [_P, _P, C.c_int32, C.POINTER(C.c_int32)]),
    'avae_debug_stamps': (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    'avae_debug_present_ids': (C.c_int, [_P, C.POINTER(C.c_int32)]),
    'avae_debug_team_batch': (C.c_int, [C.c_int32]),
    'avae_bucket_count': (C.c_int, [_P]),
    'avae_bucket_info': (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
}

_lib = None


def build(force=False, jobs=4):
    """compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(['make', '-C', CSRC, 'clean'], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(['make', '-C', CSRC, '-j%d' % jobs], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def load():
    """-> ctypes.CDLL with every entry point typed.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libargsim_vae.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the HIP path)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
