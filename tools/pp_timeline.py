"""Timeline of one workgroup of the ping-pong dK/dV kernel (PLM_ATTN_PP=2 debug instantiation): cycle stamps of waves 0 and 4."""
import ctypes, os, sys
os.environ['PLM_ATTN_PP'] = '2'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plainlm_amd import ops, _lib

B, T, nh = 32, 1024, 12
dev = 'cuda'
torch.manual_seed(0)
qkv = (torch.randn(B * T, 3 * nh * 64, device=dev) * 0.5).bfloat16()
cos = torch.rand(T, 32, device=dev); sin = torch.rand(T, 32, device=dev)
out, lse = ops.attn_fwd(qkv, B, T, nh)
dout = torch.randn(B * T, nh * 64, device=dev).bfloat16()
for _ in range(3):
  ops.attn_bwd(qkv, out, dout, lse, cos, sin, B, T, nh)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 512)()
f = lib.plm_debug_pp_timeline
f.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
print('rc', f(buf))
t0 = min(buf[w * 64] for w in range(8))
for w in range(8):
  tl = list(buf[w * 64:(w + 1) * 64])
  print('wave', w, 'start', tl[0] - t0, 'kv loaded', tl[1] - t0, ' prologue barrier', tl[2] - t0, ' loop end', tl[3] - t0, ' kernel end', tl[4] - t0)
for step in range(0, 14):
  if not any(buf[w * 64 + 8 + 4 * step] for w in range(8)):
    continue
  print('step', step)
  for w in range(8):
    e = [buf[w * 64 + 8 + 4 * step + i] for i in range(4)]
    if not any(e):
      continue
    print('   wave %d: begin %6d  end-of-first-phase %6d  after-barrier %6d  end-of-second-phase %6d' % ((w,) + tuple(x - t0 for x in e)))
