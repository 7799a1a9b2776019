"""argsim_amd -- MI355X-native (gfx950) implementation of the argsim/argsim sequence-VAE hot path.

Layout:  csrc/ (HIP kernels + C ABI, built into libargsim_vae.so), lib.py (ctypes binding),
model.py (host mirror of the reference's vAe Record), train.py / util_*.py (host loop, batching,
tokenisation), dist.py (data-parallel gradient all-reduce over RCCL), synth.py (synthetic batches).
"""
__all__ = ['lib', 'model', 'synth']
