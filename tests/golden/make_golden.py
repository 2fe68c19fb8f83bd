"""Generate golden vectors by IMPORTING the reference (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference's Python cannot travel to the GPU box, so its outputs on small
seeded inputs are frozen here as ``.npz`` data files (inputs + expected
outputs only; no reference source text).  ``wandb`` and ``absl`` are absent in
this image; ``models/construct.py:2`` and ``utils.py:11`` import them at module
top, so inert stand-in modules are registered before the import.

The script also prints the oracle-vs-reference error for every vector so a
drift is visible at generation time; ``tests/test_oracle_golden.py`` re-checks
the oracle against the committed files on every CPU test run.
"""

import os
import sys
import types
from collections import namedtuple

os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('PLAINLM_REFERENCE', '/root/reference')


def _install_stubs():
  wandb = types.ModuleType('wandb')
  wandb.run = None
  wandb.log = lambda *a, **k: None
  wandb.init = lambda *a, **k: None
  sys.modules['wandb'] = wandb
  absl = types.ModuleType('absl')
  app = types.ModuleType('absl.app')
  flags = types.ModuleType('absl.flags')
  app.run = lambda f: None

  class _Flags:
    job_idx = None
    config = None
    job_cluster = None

  flags.FLAGS = _Flags()
  for n in ('DEFINE_string', 'DEFINE_integer', 'DEFINE_bool', 'DEFINE_float'):
    setattr(flags, n, lambda *a, **k: None)
  absl.app, absl.flags = app, flags
  sys.modules.update({'absl': absl, 'absl.app': app, 'absl.flags': flags})


def main():
  _install_stubs()
  sys.path.insert(0, REF)
  sys.path.insert(1, ROOT)
  import torch.nn.functional as F
  from models.components import RMSNorm, GLU
  from models.embeddings import precompute_freqs_cis, apply_rotary_emb_complex_like
  from models.transformer import Transformer, ModelConfig
  from models import construct_model
  from engine.engine import TorchEngine
  from data.datasets.data_prep_utils import intra_doc_causal_mask
  from oracle import cpu_ref as O

  torch.manual_seed(1234)
  rng = np.random.default_rng(1234)
  ops = {}

  def report(name, a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    print(f'  {name:28s} max|oracle-ref| = {err:.3e}   (max|ref| = {ref:.3e})')

  # ---- RMSNorm fwd/bwd (models/components.py:16-28) --------------------
  M, d = 48, 128
  x = torch.randn(M, d, requires_grad=True)
  mod = RMSNorm(d, 1e-6)
  with torch.no_grad():
    mod.weight.copy_(1 + 0.1 * torch.randn(d))
  dy = torch.randn(M, d)
  y = mod(x)
  y.backward(dy)
  ops.update(rms_x=x.detach(), rms_w=mod.weight.detach(), rms_y=y.detach(), rms_dy=dy,
             rms_dx=x.grad, rms_dw=mod.weight.grad)
  report('rmsnorm fwd', O.rmsnorm(x.detach(), mod.weight.detach()), y.detach())

  # ---- RoPE (models/embeddings.py:8-30) --------------------------------
  B, T, nh, hd = 2, 64, 2, 64
  freqs = precompute_freqs_cis(hd, 2048, 500000)
  q = torch.randn(B, T, nh, hd)
  k = torch.randn(B, T, nh, hd)
  qr, kr = apply_rotary_emb_complex_like(q, k, freqs_cis=freqs[:, :T])
  rows = [0, 1, 2, 63, 64, 1023, 1024, 2047]
  ops.update(rope_q=q, rope_k=k, rope_qr=qr.contiguous(), rope_kr=kr.contiguous(),
             rope_rows=np.array(rows), rope_cos_rows=freqs[0, rows, 0, :, 0], rope_sin_rows=freqs[0, rows, 0, :, 1])
  cos, sin = O.rope_table(hd, 2048)
  report('rope table cos', cos[rows], freqs[0, rows, 0, :, 0])
  report('rope table sin', sin[rows], freqs[0, rows, 0, :, 1])
  report('rope apply q', O.rope_apply(q, cos, sin), qr)
  report('rope apply k', O.rope_apply(k, cos, sin), kr)

  # ---- SDPA causal + doc-masked (models/transformer.py:49-65) ----------
  def sdpa(qq, kk, vv, mask):
    qh, kh, vh = (t.transpose(1, 2) for t in (qq, kk, vv))
    if mask is None:
      o = F.scaled_dot_product_attention(qh, kh, vh, is_causal=True)
    else:
      o = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=mask.unsqueeze(1))
    return o.transpose(1, 2).contiguous().view(qq.shape[0], qq.shape[1], -1)

  docs_lengths = [[10, 20, 35], [1, 31, 32, 1]]  # each sums to T+1 = 65
  masks = torch.stack([intra_doc_causal_mask(dl, T + 1) for dl in docs_lengths])[:, :T, :T].contiguous()
  for tag, mask in (('c', None), ('m', masks)):
    qq, kk, vv = (torch.randn(B, T, nh, hd, requires_grad=True) for _ in range(3))
    do = torch.randn(B, T, nh * hd)
    o = sdpa(qq, kk, vv, mask)
    o.backward(do)
    ops.update({f'att{tag}_q': qq.detach(), f'att{tag}_k': kk.detach(), f'att{tag}_v': vv.detach(),
                f'att{tag}_o': o.detach(), f'att{tag}_do': do, f'att{tag}_dq': qq.grad,
                f'att{tag}_dk': kk.grad, f'att{tag}_dv': vv.grad})
    ds = None if mask is None else O.doc_start_from_lengths(docs_lengths, T)
    report(f'attention {tag}', O.attention(qq.detach(), kk.detach(), vv.detach(), ds), o.detach())
  ops['attm_docs_lengths'] = np.array([dl + [0] * (8 - len(dl)) for dl in docs_lengths])
  ops['attm_mask'] = masks
  report('doc mask', O.mask_from_doc_start(O.doc_start_from_lengths(docs_lengths, T)).float(), masks.float())

  # ---- GLU (models/components.py:43-56) ---------------------------------
  hid_cases = [(128, 8 / 3), (768, 8 / 3), (1024, 8 / 3), (64, 8 / 3), (768, 4.0)]
  hid = []
  for dd, ex in hid_cases:
    g = GLU(dd, int(ex * dd))
    hid.append(g.hidden_dim)
    assert O.glu_hidden_dim(dd, ex) == g.hidden_dim, (dd, ex)
  ops['hid_dim'] = np.array([c[0] for c in hid_cases])
  ops['hid_expand'] = np.array([c[1] for c in hid_cases])
  ops['hid_out'] = np.array(hid)
  glu = GLU(128, int(8 / 3 * 128))
  xg = torch.randn(1, M, 128, requires_grad=True)  # reference GLU splits on dim=2
  dyg = torch.randn(1, M, 128)
  yg = glu(xg)
  yg.backward(dyg)
  ops.update(glu_x=xg.detach(), glu_w1=glu.fc1.weight.detach(), glu_w2=glu.fc2.weight.detach(), glu_y=yg.detach(),
             glu_dy=dyg, glu_dx=xg.grad, glu_dw1=glu.fc1.weight.grad, glu_dw2=glu.fc2.weight.grad)
  report('glu fwd', O.swiglu(xg.detach() @ glu.fc1.weight.detach().t(), glu.hidden_dim) @ glu.fc2.weight.detach().t(),
         yg.detach())

  # ---- CrossEntropy (engine/engine.py:81,111) ----------------------------
  V = 777
  lg = (3 * torch.randn(M, V)).requires_grad_(True)
  tg = torch.from_numpy(rng.integers(0, V, size=M))
  loss = torch.nn.CrossEntropyLoss()(lg, tg)
  loss.backward()
  ops.update(ce_logits=lg.detach(), ce_targets=tg, ce_loss=loss.detach(), ce_dlogits=lg.grad)
  report('cross entropy', O.cross_entropy(lg.detach(), tg), loss.detach())

  # ---- Embedding bwd with repeated ids (models/transformer.py:94,110) ----
  emb = torch.nn.Embedding(50, 32)
  ids = torch.from_numpy(rng.integers(0, 12, size=(4, 16)))
  eo = emb(ids)
  deo = torch.randn_like(eo)
  eo.backward(deo)
  ops.update(emb_w=emb.weight.detach(), emb_ids=ids, emb_out=eo.detach(), emb_dout=deo, emb_dw=emb.weight.grad)

  np.savez_compressed(os.path.join(HERE, 'ops.npz'), **{k: np.asarray(v) for k, v in ops.items()})

  # ---- whole model fwd/bwd (models/transformer.py:86-140) ----------------
  Cfg = dict(vocab_size=256, seq_len=64, dim=128, expand=8 / 3, n_layers=2, n_heads=2, mlp='glu')
  torch.manual_seed(100)
  model = Transformer(ModelConfig(**Cfg))
  names = [n for n, _ in model.named_parameters()]
  ocfg = O.OracleConfig(vocab_size=256, seq_len=64, dim=128, n_layers=2, n_heads=2)
  assert names == O.param_names(ocfg), names
  assert {n: tuple(p.shape) for n, p in model.named_parameters()} == O.param_shapes(ocfg)
  stats = {n: (p.detach().mean().item(), p.detach().std().item() if p.numel() > 1 else 0.0) for n, p in model.named_parameters()}
  weights = {n: p.detach().clone() for n, p in model.named_parameters()}
  tok = torch.from_numpy(rng.integers(0, 256, size=(2, 65)))
  ids, tgt = tok[:, :64], tok[:, 1:65]
  logits = model(ids, None)
  loss = torch.nn.CrossEntropyLoss()(logits.view(-1, 256), tgt.reshape(-1))
  loss.backward()
  grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
  with torch.no_grad():
    logits_m = model(ids, masks)
  mdl = {'tokens': tok, 'logits': logits.detach(), 'loss': loss.detach(), 'logits_docmask': logits_m,
         'docs_lengths': ops['attm_docs_lengths'],
         'init_mean': np.array([stats[n][0] for n in names]), 'init_std': np.array([stats[n][1] for n in names])}
  mdl.update({'w:' + n: w for n, w in weights.items()})
  mdl.update({'g:' + n: g for n, g in grads.items()})
  np.savez_compressed(os.path.join(HERE, 'model.npz'), **{k: np.asarray(v) for k, v in mdl.items()})
  ol, og = O.loss_and_grads(weights, ocfg, ids, tgt)
  report('model logits', O.forward(weights, ocfg, ids), logits.detach())
  report('model logits docmask', O.forward(weights, ocfg, ids, O.doc_start_from_lengths(docs_lengths, 64)), logits_m)
  report('model loss', ol, loss.detach())
  for n in names:
    report('grad ' + n, og[n], grads[n])

  # ---- engine: loss sequence + params over optimizer steps (engine/engine.py:93-141)
  EC = dict(model='transformer', vocab_size=256, seq_len=64, d_model=128, expand='8/3', n_layers=2, n_heads=2,
            mlp_class='glu', tie_embeddings=False, torch_compile=False, micro_batch_size=1,
            grad_accumulation_steps=4, dtype='bfloat16', optim='adamw', fused_optim=False, lr=3e-3,
            weight_decay=0.1, beta1=0.9, beta2=0.95, grad_clip=1.0, scheduler='warmup_cosine', warmup_steps=2,
            cooldown_steps=None, lr_start=0.0, lr_end=1e-5, lr_end_pct=None, steps_budget=8, resume=False, seed=100)
  ecfg = namedtuple('Config', EC.keys())(**EC)
  torch.manual_seed(100)
  emodel, _ = construct_model(ecfg)
  emodel.load_state_dict(weights)
  eng = TorchEngine(emodel, ecfg, 'cpu', None, None)
  n_opt = 4
  etok = torch.from_numpy(rng.integers(0, 256, size=(n_opt * 4, 1, 65)))
  losses, lrs, norms = [], [], []
  for i in range(n_opt * 4):
    losses.append(eng.step({'input_ids': etok[i]}).item())
    if (i + 1) % 4 == 0:
      lrs.append(eng.optimizer.param_groups[0]['lr'])
      norms.append([p.detach().norm().item() for _, p in emodel.named_parameters()])
  final = {n: p.detach().clone() for n, p in emodel.named_parameters()}
  en = {'tokens': etok, 'losses': np.array(losses), 'lrs': np.array(lrs), 'param_norms': np.array(norms),
        'final:layers.1.mlp.fc2.weight': final['layers.1.mlp.fc2.weight'], 'final:out_norm.weight': final['out_norm.weight'],
        'final:layers.0.attn_norm.weight': final['layers.0.attn_norm.weight'],
        'final:embed_rows': final['embed_tokens.weight'][:16]}
  np.savez_compressed(os.path.join(HERE, 'engine.npz'), **{k: np.asarray(v) for k, v in en.items()})
  oe = O.OracleEngine(weights, ocfg, lr=3e-3, weight_decay=0.1, beta1=0.9, beta2=0.95, grad_clip=1.0, accum=4,
                      steps_budget=8, warmup_steps=2)
  ol = [oe.step({'input_ids': etok[i]}).item() for i in range(n_opt * 4)]
  report('engine losses', np.array(ol), np.array(losses))
  report('engine final fc2', oe.params['layers.1.mlp.fc2.weight'], final['layers.1.mlp.fc2.weight'])
  report('engine final out_norm', oe.params['out_norm.weight'], final['out_norm.weight'])
  print('losses', np.round(losses, 5))
  print('lrs', lrs)
  make_checkpoint_fixture(construct_model, TorchEngine, rng)
  for f in ('ops.npz', 'model.npz', 'engine.npz', 'ckpt.npz', 'ref_ckpt_step_2.pth'):
    print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'KiB')


CKPT_CFG = dict(model='transformer', vocab_size=128, seq_len=32, d_model=64, expand='8/3', n_layers=2, n_heads=1,
                mlp_class='glu', tie_embeddings=False, torch_compile=False, micro_batch_size=1,
                grad_accumulation_steps=2, dtype='bfloat16', optim='adamw', fused_optim=False, lr=3e-3,
                weight_decay=0.1, beta1=0.9, beta2=0.95, grad_clip=1.0, scheduler='warmup_cosine', warmup_steps=2,
                cooldown_steps=None, lr_start=0.0, lr_end=1e-5, lr_end_pct=None, steps_budget=8, resume=False, seed=100)


def make_checkpoint_fixture(construct_model, TorchEngine, rng):
  """a checkpoint WRITTEN BY THE REFERENCE engine (the dict of checkpoint_utils.py:32-38, saved with
  torch.save like :45) after 2 optimizer steps of a small model, plus what the reference does next: the losses of the
  following 2 optimizer steps when it resumes from that file (engine/engine.py:56-60,86-89).  ckpt.npz also carries the
  initial weights and every batch, so the GPU test can run the first two steps itself and write ITS checkpoint."""
  ecfg = namedtuple('Config', CKPT_CFG.keys())(**CKPT_CFG)
  torch.manual_seed(100)
  model, _ = construct_model(ecfg)
  init = {n: p.detach().clone() for n, p in model.named_parameters()}
  eng = TorchEngine(model, ecfg, 'cpu', None, None)
  tok = torch.from_numpy(rng.integers(0, CKPT_CFG['vocab_size'], size=(8, 1, CKPT_CFG['seq_len'] + 1)))
  losses = [eng.step({'input_ids': tok[i]}).item() for i in range(4)]
  state = {'step': 2, 'state_dict': model.state_dict(), 'optimizer': eng.optimizer.state_dict(),
           'scheduler': eng.scheduler.state_dict(), 'scaler': eng.scaler.state_dict()}
  path = os.path.join(HERE, 'ref_ckpt_step_2.pth')
  torch.save(state, path)
  # resume exactly as train.py does: fresh model, cfg.resume = True, checkpoint loaded on the CPU
  rcfg = namedtuple('Config', CKPT_CFG.keys())(**dict(CKPT_CFG, resume=True))
  model2, _ = construct_model(rcfg)
  eng2 = TorchEngine(model2, rcfg, 'cpu', None, torch.load(path, map_location='cpu', weights_only=False))
  assert eng2.micro_steps == 4
  resumed = [eng2.step({'input_ids': tok[i]}).item() for i in range(4, 8)]
  cont = [eng.step({'input_ids': tok[i]}).item() for i in range(4, 8)]  # the uninterrupted run, for reference
  assert np.allclose(resumed, cont, rtol=1e-6), (resumed, cont)
  out = {'tokens': tok, 'losses': np.array(losses + resumed), 'lr_after': np.array(eng2.optimizer.param_groups[0]['lr']),
         'sched_iter': np.array(eng2.scheduler.iter)}
  out.update({'w:' + n: v for n, v in init.items()})
  out.update({'final:' + n: p.detach().clone() for n, p in model2.named_parameters() if 'norm' in n or n.endswith('w_out.weight')})
  np.savez_compressed(os.path.join(HERE, 'ckpt.npz'), **{k: np.asarray(v) for k, v in out.items()})
  print('checkpoint fixture: losses', np.round(losses + resumed, 5), 'scheduler', eng2.scheduler.state_dict())


if __name__ == '__main__':
  main()
