"""ctypes binding of libplainlm_hip.so (the C ABI declared in include/plainlm_hip.h).

There is NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C plainlm_amd/csrc``.
"""

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libplainlm_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'plainlm_hip.h')

_lib = None

# ABI the signatures below were written for (plm_version() of the library must match: a stale .so that still exports every
# symbol but with other argument lists / struct layouts would corrupt memory instead of raising)
EXPECTED_ABI = 107

_P = C.c_void_p
_I64 = C.c_int64
_I = C.c_int
_F = C.c_float
_SZ = C.c_size_t

class CastItem(C.Structure):
  """struct plm_cast_item (include/plainlm_hip.h)."""
  _fields_ = [('src', _P), ('dst', _P), ('dst_t', _P), ('rows', _I64), ('cols', _I64), ('ld_t', _I64)]


class AdamwItem(C.Structure):
  """struct plm_adamw_item (include/plainlm_hip.h)."""
  _fields_ = [('p', _P), ('g', _P), ('m', _P), ('v', _P), ('dst', _P), ('dst_t', _P), ('rows', _I64), ('cols', _I64), ('ld_t', _I64)]


class ColsumItem(C.Structure):
  """struct plm_colsum_item (include/plainlm_hip.h)."""
  _fields_ = [('part', _P), ('out', _P), ('accumulate', _I)]


class TnProblem(C.Structure):
  """struct plm_tn_problem (include/plainlm_hip.h)."""
  _fields_ = [('A', _P), ('lda', _I64), ('B', _P), ('ldb', _I64), ('C', _P), ('ldc', _I64), ('M', _I64), ('N', _I64),
              ('accumulate', _I), ('alpha_dev', _P)]


# name -> (restype, argtypes); must cover every function in include/plainlm_hip.h
SIGNATURES = {
  'plm_version': (_I, []),
  'plm_last_error_string': (C.c_char_p, []),
  'plm_reload_env': (None, []),
  'plm_cast_f32_bf16': (_I, [_P, _P, _I64, _P]),
  'plm_cast_f32_bf16_t': (_I, [_P, _P, _P, _I64, _I64, _I64, _P]),
  'plm_cast_f32_bf16_t_multi': (_I, [C.POINTER(CastItem), _I, _P]),
  'plm_embed_fwd': (_I, [_P, _P, _P, _I64, _I64, _I64, _P]),
  'plm_embed_bwd': (_I, [_P, _P, _P, _I64, _I64, _I64, _P]),
  'plm_embed_bwd_workspace_bytes': (_SZ, [_I64, _I64]),
  'plm_embed_bwd_sorted': (_I, [_P, _P, _P, _I64, _I64, _I64, _I, _P, _SZ, _P]),
  'plm_rmsnorm_fwd': (_I, [_P, _P, _P, _P, _P, _P, _I64, _I64, _F, _P]),
  'plm_rmsnorm_bwd_blocks': (_I64, [_I64]),
  'plm_rmsnorm_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P]),
  'plm_colsum_f32': (_I, [_P, _P, _I64, _I64, _I, _P]),
  'plm_colsum_f32_multi': (_I, [C.POINTER(ColsumItem), _I, _I64, _I64, _P]),
  'plm_swiglu_fwd': (_I, [_P, _P, _I64, _I64, _P]),
  'plm_swiglu_bwd': (_I, [_P, _P, _P, _I64, _I64, _P]),
  'plm_act_fwd': (_I, [_P, _P, _I64, _I, _P]),
  'plm_act_bwd': (_I, [_P, _P, _P, _I64, _I, _P]),
  'plm_gemm_bf16_nt': (_I, [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _I, _I, _P, _P]),
  'plm_gemm_bf16_nt_ex': (_I, [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _I, _I, _P, _I, _P]),
  'plm_gemm_nt_workspace_bytes': (_SZ, [_I64, _I64, _I64]),
  'plm_gemm_bf16_nt_ws': (_I, [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _I, _I, _P, _I, _P, _SZ, _P]),
  'plm_gemm_tn_workspace_bytes': (_SZ, [_I64, _I64, _I64]),
  'plm_gemm_bf16_tn': (_I, [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _I, _P, _P, _SZ, _P]),
  'plm_gemm_tn_grouped_workspace_bytes': (_SZ, [C.POINTER(_I64), C.POINTER(_I64), _I, _I64]),
  'plm_gemm_bf16_tn_grouped': (_I, [C.POINTER(TnProblem), _I, _I64, _P, _SZ, _P]),
  'plm_rope_qk': (_I, [_P, _P, _P, _I64, _I64, _I64, _I64, _P]),
  'plm_qkv_rope_bf16': (_I, [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _P, _P, _I64, _I64, _I64, _I64, _P]),
  'plm_fc1_swiglu_bf16': (_I, [_P, _I64, _P, _I64, _P, _P, _I64, _I64, _I64, _P]),
  'plm_fc2_dx_swiglu_bwd_bf16': (_I, [_P, _I64, _P, _I64, _P, _P, _P, _I64, _I64, _I64, _P]),
  'plm_attn_doc_start_from_mask': (_I, [_P, _P, _P, _I64, _I64, _P]),
  'plm_attn_doc_plan_bytes': (_I64, [_I64, _I64]),
  'plm_attn_doc_plan': (_I, [_P, _P, _I64, _I64, _I64, _P]),
  'plm_attn_fwd': (_I, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P]),
  'plm_attn_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P]),
  'plm_ce_fwd_bwd': (_I, [_P, _P, _P, _I64, _I64, _I64, _F, _P]),
  'plm_mean_f32': (_I, [_P, _P, _I64, _P]),
  'plm_scale_bf16': (_I, [_P, _I64, _P, _P]),
  'plm_axpy_f32': (_I, [_P, _P, _I64, _P, _I, _P]),
  'plm_sumsq_f32': (_I, [_P, _I64, _P, _P, _P]),
  'plm_adamw_f32': (_I, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _F, _F, _P, _P]),
  'plm_adamw_cast_multi': (_I, [C.POINTER(AdamwItem), _I, _F, _F, _F, _F, _F, _F, _F, _P, _P]),
  'plm_set_cu_reserve': (_I, [_I]),
  'plm_comm_unique_id': (_I, [_P]),
  'plm_comm_init': (_I, [C.POINTER(_P), _P, _I, _I, _I]),
  'plm_comm_init_capped': (_I, [C.POINTER(_P), _P, _I, _I, _I, _I]),
  'plm_comm_split': (_I, [_P, C.POINTER(_P), _I]),
  'plm_comm_destroy': (_I, [_P]),
  'plm_comm_allreduce_avg_f32': (_I, [_P, _P, _I64, _P]),
  'plm_comm_rsag_avg_f32': (_I, [_P, _P, _I64, _P]),
  'plm_comm_broadcast_f32': (_I, [_P, _P, _I64, _I, _P]),
  'plm_probe_ds_read_tr16': (_I, [_P, _P]),
  'plm_probe_mfma32': (_I, [_P, _P, _P, _P]),
}


def header_functions():
  """Names of all functions declared in include/plainlm_hip.h."""
  with open(HEADER_PATH) as f:
    src = re.sub(r'/\*.*?\*/', '', f.read(), flags=re.S)
  return sorted(set(re.findall(r'\b(plm_[a-z0-9_]+)\s*\(', src)))


def load():
  """Load the library (once).  Raises RuntimeError when it is absent — by design there is no CPU fallback."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise RuntimeError(
      f'{LIB_PATH} not found: the HIP extension is not built. Run `make -C plainlm_amd/csrc` '
      '(or __graft_entry__.build()). plainlm_amd has no CPU fallback.')
  lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
  for name, (res, args) in SIGNATURES.items():
    fn = getattr(lib, name)  # AttributeError -> symbol missing, loud by construction
    fn.restype = res
    fn.argtypes = args
  got = lib.plm_version()
  if got != EXPECTED_ABI:
    raise RuntimeError(f'{LIB_PATH} reports ABI version {got}, this package was written for {EXPECTED_ABI}: '
                       'rebuild it (make -C plainlm_amd/csrc)')
  _lib = lib
  return lib


def check(rc, what):
  if rc != 0:
    msg = load().plm_last_error_string()
    raise RuntimeError(f'{what} failed (rc={rc}): {msg.decode() if msg else "?"}')
