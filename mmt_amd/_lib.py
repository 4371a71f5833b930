"""ctypes binding of libmmt_hip.so (the C ABI declared in include/mmt_hip.h).

There is NO fallback: if the library is missing the import of any product module fails loudly, and
calling a kernel without a GPU raises.  The oracle is never imported from here.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libmmt_hip.so')

c_int, c_i64, c_u32, c_f32, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_uint32, ctypes.c_float, ctypes.c_void_p


class MmtEpilogue(ctypes.Structure):
  _fields_ = [('bias', c_vp), ('res', c_vp), ('ldres', c_i64), ('out2', c_vp), ('ldout2', c_i64),
              ('aux', c_vp), ('ldaux', c_i64), ('colsum', c_vp), ('row_index', c_vp),
              ('drop_key', c_u32), ('drop_thr16', c_u32), ('drop_scale', c_f32), ('reserved', ctypes.c_int32)]


EPI = dict(BF16=0, BIAS_BF16=1, BIAS_GELU=2, BIAS_DROP_RES=3, DGELU=4, ADD_F32=5, F32=6, BIAS_F32=7)

# name -> (restype, argtypes); must list every symbol declared in include/mmt_hip.h
SIGNATURES = {
    'mmt_abi_version': (c_int, []),
    'mmt_build_info': (ctypes.c_char_p, []),
    'mmt_gemm_nt_bf16': (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_int,
                                 ctypes.POINTER(MmtEpilogue), c_vp, c_vp]),
    'mmt_gemm_tn_bf16': (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    'mmt_reduce_slabs': (c_int, [c_vp, c_int, c_i64, c_vp, c_int, c_vp]),
    'mmt_ln_fwd': (c_int, [c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    'mmt_embed_ln_fwd': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp,
                                 c_int, c_int, c_vp, c_vp, c_u32, c_u32, c_f32, c_vp]),
    'mmt_ln_bwd_rows_per_block': (c_int, []),
    'mmt_ln_bwd': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp,
                           c_u32, c_u32, c_f32, c_vp]),
    'mmt_col_reduce': (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_vp]),
    'mmt_table_grad': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_int, c_vp]),
    'mmt_attn_fwd': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_f32, c_u32, c_u32,
                             c_f32, c_vp]),
    'mmt_attn_bwd': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_f32,
                             c_u32, c_u32, c_f32, c_vp]),
    'mmt_attn_dropout_mask': (c_int, [c_vp, c_int, c_int, c_int, c_u32, c_u32, c_vp]),
}

_lib = None


def lib():
  """Loads the shared library (once).  Raises if it has not been built."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise RuntimeError(
          'libmmt_hip.so is missing (%s): run `python -m mmt_amd.build` (hipcc, gfx950). '
          'There is no CPU fallback for the MMT hot path.' % LIB_PATH)
    handle = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
      fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
      fn.restype = res
      fn.argtypes = args
    if handle.mmt_abi_version() != 1:
      raise RuntimeError('libmmt_hip.so ABI mismatch')
    _lib = handle
  return _lib


def check(rc, what):
  if rc != 0:
    raise RuntimeError('%s failed with code %d' % (what, rc))
