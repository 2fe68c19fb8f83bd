"""Where does the engine's trajectory leave the oracle's?  (GPU box; diagnostic, not a test.)

Runs the 16-micro-step engine fixture (tests/golden/engine.npz) on HipEngine and, in lock step, on the bf16-emulating
oracle engine (oracle/cpu_ref_bf16.py) and the fp32 oracle engine, and prints per optimizer window
  * every parameter's accumulated gradient: GPU vs bf16-emulation vs fp32 (max error relative to the tensor's max),
  * every parameter after the update: |GPU - emulation| in units of the learning rate,
  * the loss of every micro-step against both oracles and against the reference's own golden sequence.
"""
import os
import sys
from collections import namedtuple

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cpu_ref as O  # noqa: E402
from oracle import cpu_ref_bf16 as E  # noqa: E402
import plainlm_amd as P  # noqa: E402


def relmax(a, ref):
  a, ref = a.double().cpu(), ref.double().cpu()
  return ((a - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def main():
  z = np.load(os.path.join(ROOT, 'tests', 'golden', 'model.npz'))
  w = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('w:')}
  en = np.load(os.path.join(ROOT, 'tests', 'golden', 'engine.npz'))
  tokens = torch.from_numpy(en['tokens'])
  accum = int(os.environ.get('ACCUM', '4'))
  EC = dict(model='transformer', vocab_size=256, seq_len=64, d_model=128, expand='8/3', n_layers=2, n_heads=2,
            mlp_class='glu', tie_embeddings=False, torch_compile=False, micro_batch_size=1, grad_accumulation_steps=accum,
            dtype='bfloat16', optim='adamw', fused_optim=True, lr=3e-3, weight_decay=0.1, beta1=0.9, beta2=0.95,
            grad_clip=1.0, scheduler='warmup_cosine', warmup_steps=2, cooldown_steps=None, lr_start=0.0, lr_end=1e-5,
            lr_end_pct=None, steps_budget=8, resume=False, seed=100)
  cfg = namedtuple('Config', EC.keys())(**EC)
  model, _ = P.construct_model(cfg)
  model.load_state_dict(w)
  eng = P.TorchEngine(model, cfg, 'cuda', None, None)
  ocfg = O.OracleConfig(vocab_size=256, seq_len=64, dim=128, n_layers=2, n_heads=2)
  kw = dict(lr=3e-3, weight_decay=0.1, beta1=0.9, beta2=0.95, grad_clip=1.0, accum=accum, steps_budget=8, warmup_steps=2)
  emu = E.OracleEngineBF16(w, ocfg, **kw)
  f32 = O.OracleEngine(w, ocfg, **kw)
  names = [n for n, _ in eng.model.named_parameters()]

  # kernel paths at this micro-batch shape (B = 1: M = 64 token rows), without the engine: one fwd+bwd per batch size
  for B in (1, 2):
    m2 = P.Transformer(P.ModelConfig(vocab_size=256, seq_len=64, dim=128, expand=8 / 3, n_layers=2, n_heads=2, mlp='glu'))
    m2.load_state_dict(w)
    m2 = m2.cuda()
    m2.enable_main_grad()
    tk = tokens[:B, 0]
    ids, tgt = tk[:, :64], tk[:, 1:65]
    m2.sink.begin_window()
    m2.loss(ids.cuda(), tgt.cuda()).backward()
    m2.sink.flush_dw()
    torch.cuda.synchronize()
    _, ge = E.loss_and_grads(w, ocfg, ids, tgt)
    _, gf = O.loss_and_grads(w, ocfg, ids, tgt)
    print(f'plain model, B={B}: gradient error (rel. to max)   gpu-vs-emu   gpu-vs-fp32   emu-vs-fp32')
    for n, p_ in m2.named_parameters():
      g = p_.main_grad.float().cpu()
      print(f'    {n:32s} {relmax(g, ge[n]):10.2e}  {relmax(g, gf[n]):10.2e}  {relmax(ge[n], gf[n]):10.2e}')

  snap = {}
  orig = eng.optimizer.clip_and_step

  def spy(max_norm=None):
    eng.model.sink.flush_dw()
    torch.cuda.synchronize()
    snap['g'] = {n: p.main_grad.detach().float().cpu().clone() for n, p in eng.model.named_parameters()}
    return orig(max_norm)

  eng.optimizer.clip_and_step = spy
  for i in range(tokens.shape[0]):
    b = {'input_ids': tokens[i]}
    eg = emu.grads  # accumulated so far (before this micro-step)
    lg = eng.step(b).item()
    # oracle engines: capture the accumulated gradient before their optimizer step consumes it
    last = (i + 1) % accum == 0
    if last:
      for o_ in (emu, f32):
        o_._keep = None
        orig_opt = o_._optimizer_step

        def wrapped(o_=o_, orig_opt=orig_opt):
          o_._keep = {n: t.clone() for n, t in o_.grads.items()}
          orig_opt()

        o_._optimizer_step = wrapped
    le, lf = emu.step(b).item(), f32.step(b).item()
    ref = en['losses'][i] if accum == 4 else float('nan')
    print(f'micro {i + 1:2d}  loss gpu {lg:.6f}  emu {le:.6f}  fp32 {lf:.6f}  ref {ref:.6f}   |gpu-emu|/emu {abs(lg - le) / le:.2e}   '
          f'|gpu-fp32|/fp32 {abs(lg - lf) / lf:.2e}   |emu-fp32| {abs(le - lf) / lf:.2e}')
    if last:
      torch.cuda.synchronize()
      lr = emu.lr
      print(f'  window {(i + 1) // accum}: gradient error (rel. to max)      gpu-vs-emu   gpu-vs-fp32   emu-vs-fp32    | param |gpu-emu| max, in units of lr={kw["lr"]:g}')
      for n in names:
        g = snap['g'][n]
        p_gpu = dict(eng.model.named_parameters())[n].detach().float().cpu()
        print(f'    {n:32s} {relmax(g, emu._keep[n]):10.2e}  {relmax(g, f32._keep[n]):10.2e}  {relmax(emu._keep[n], f32._keep[n]):10.2e}    |'
              f' {(p_gpu - emu.params[n]).abs().max().item() / kw["lr"]:9.2e}')
      for o_ in (emu, f32):
        del o_._optimizer_step


if __name__ == '__main__':
  main()
