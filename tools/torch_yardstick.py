"""Yardstick only (not the product path, not the oracle): the same 160M decoder written with stock PyTorch modules - nn.Linear,
F.scaled_dot_product_attention(is_causal=True), F.silu, F.cross_entropy under bf16 autocast with fp32 parameters, the way the
reference runs on a GPU - timed on the same MI355X for fwd+bwd tokens/s.  It answers "what would stock
PyTorch-ROCm give on this box", next to bench.py's number for the hand-written path.
Usage: python tools/torch_yardstick.py [--steps 10] [--compile]"""
import argparse
import json
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class RMSNorm(nn.Module):
  def __init__(self, d, eps=1e-6):
    super().__init__()
    self.eps, self.weight = eps, nn.Parameter(torch.ones(d))

  def forward(self, x):
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.eps)).type_as(x) * self.weight


def rope(x, cos, sin):  # interleaved pairs, x [B, T, nh, hd]
  a, b = x.float()[..., 0::2], x.float()[..., 1::2]
  c, s = cos[None, :, None, :], sin[None, :, None, :]
  return torch.stack((a * c - b * s, b * c + a * s), dim=-1).flatten(-2).type_as(x)


class Block(nn.Module):
  def __init__(self, d, nh, h):
    super().__init__()
    self.nh = nh
    self.attn_norm, self.mlp_norm = RMSNorm(d), RMSNorm(d)
    self.w_qkv, self.w_out = nn.Linear(d, 3 * d, bias=False), nn.Linear(d, d, bias=False)
    self.fc1, self.fc2 = nn.Linear(d, 2 * h, bias=False), nn.Linear(h, d, bias=False)

  def forward(self, x, cos, sin):
    B, T, d = x.shape
    q, k, v = self.w_qkv(self.attn_norm(x)).view(B, T, 3, self.nh, d // self.nh).unbind(2)
    q, k = rope(q, cos, sin), rope(k, cos, sin)
    o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), is_causal=True)
    x = x + self.w_out(o.transpose(1, 2).reshape(B, T, d))
    g, u = self.fc1(self.mlp_norm(x)).chunk(2, dim=-1)
    return x + self.fc2(F.silu(g) * u)


class Model(nn.Module):
  def __init__(self, V=50280, d=768, L=12, nh=12, h=2048, T=1024):
    super().__init__()
    self.embed = nn.Embedding(V, d)
    self.layers = nn.ModuleList([Block(d, nh, h) for _ in range(L)])
    self.norm, self.head = RMSNorm(d), nn.Linear(d, V, bias=False)
    hd = d // nh
    ang = torch.outer(torch.arange(T, dtype=torch.float32), 1.0 / (500000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd)))
    self.register_buffer('cos', torch.cos(ang), persistent=False)
    self.register_buffer('sin', torch.sin(ang), persistent=False)

  def forward(self, ids):
    x = self.embed(ids)
    for blk in self.layers:
      x = blk(x, self.cos, self.sin)
    return self.head(self.norm(x))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--steps', type=int, default=10)
  ap.add_argument('--batch', type=int, default=32)
  ap.add_argument('--compile', action='store_true')
  a = ap.parse_args()
  torch.manual_seed(0)
  dev = 'cuda'
  m = Model().to(dev)
  for p in m.parameters():
    if p.dim() > 1:
      nn.init.normal_(p, std=0.02)
  f = torch.compile(m) if a.compile else m
  B, T, V = a.batch, 1024, 50280
  tok = torch.randint(0, V, (4, B, T + 1), device=dev)

  def step(i):
    ids, tgt = tok[i % 4, :, :T], tok[i % 4, :, 1:]
    with torch.autocast('cuda', dtype=torch.bfloat16):
      logits = f(ids)
    loss = F.cross_entropy(logits.float().view(-1, V), tgt.reshape(-1))
    loss.backward()
    for p in m.parameters():
      p.grad = None
    return loss

  for i in range(3):
    step(i)
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for i in range(a.steps):
    loss = step(i)
  e.record()
  torch.cuda.synchronize()
  ms = s.elapsed_time(e) / a.steps
  print(json.dumps({'yardstick': 'stock PyTorch modules, bf16 autocast' + (' + torch.compile' if a.compile else ' (eager)'), 'batch': B, 'seq_len': T,
                    'ms_per_step': round(ms, 2), 'tokens_per_s': round(B * T / ms * 1e3, 1), 'loss': round(float(loss), 4),
                    'torch': torch.__version__, 'peak_mem_GB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))


if __name__ == '__main__':
  main()
