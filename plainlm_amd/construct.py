"""``construct_model`` / ``get_param_groups`` with the reference's call signatures
(models/construct.py:5-75), building the gfx950-native Transformer."""

from fractions import Fraction

from .transformer import ModelConfig, Transformer


def construct_model(cfg):
  """cfg: any object with the reference's YAML keys (model, vocab_size, d_model, expand,
  n_layers, n_heads, mlp_class, seq_len, tie_embeddings).  Returns (model, model_cfg)."""
  if cfg.model != 'transformer':
    raise NotImplementedError(
      f"model '{cfg.model}' is outside the accelerated hot path (only 'transformer'; the Pythia/HF branch is not built)")
  model_cfg = ModelConfig(
    vocab_size=cfg.vocab_size,
    dim=cfg.d_model,
    expand=float(Fraction(cfg.expand)),
    n_layers=cfg.n_layers,
    n_heads=cfg.n_heads,
    rmsnorm_eps=1e-6,
    mlp=cfg.mlp_class,
    seq_len=cfg.seq_len,
    tie_embeddings=cfg.tie_embeddings,
  )
  model = Transformer(model_cfg)
  print(f'Number of parameters: {model.count_params(non_embedding=False):_}')
  print(f'Number of non-embedding parameters: {model.count_params(non_embedding=True):_}')
  return model, model_cfg


def get_param_groups(model, weight_decay):
  """Decay group = parameters whose name contains neither 'bias' nor 'norm'
  (models/construct.py:54-58); everything else gets weight_decay 0."""
  named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
  decay, no_decay = [], []
  for n, p in named:
    skip = getattr(p, '_no_weight_decay', False) or 'bias' in n or 'norm' in n
    (no_decay if skip else decay).append(p)
  return [
    {'params': decay, 'weight_decay': weight_decay},
    {'params': no_decay, 'weight_decay': 0.0},
  ]
