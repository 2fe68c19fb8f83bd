"""The other direction of SURVEY 8f N4 (build container only: imports the reference).

tests/test_model_gpu.py::test_engine_checkpoint_has_the_reference_layout writes the checkpoint of a HipEngine run
(2 optimizer steps of the ckpt.npz fixture on an MI355X) to gpurun_out/hip_ckpt_step_2.pth together with the losses of its own
next 4 micro-steps (hip_ckpt_cont.npz).  This script resumes the REFERENCE engine (engine/engine.py:56-60,86-89, CPU fp32)
from that file, runs the same 4 micro-steps and compares; on success it installs both files under tests/golden/, where
tests/test_oracle_golden.py::test_reference_resumes_hip_checkpoint repeats the check whenever /root/reference is present.

  PYTHONDONTWRITEBYTECODE=1 python tests/golden/check_hip_ckpt_with_reference.py [dir with hip_ckpt_step_2.pth]
"""
import os
import shutil
import sys
from collections import namedtuple

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import CKPT_CFG, REF, ROOT, _install_stubs  # noqa: E402


def resume_with_reference(ckpt_path, tokens):
  """losses of micro-steps 4..7 when the reference's TorchEngine resumes from ckpt_path."""
  _install_stubs()
  if REF not in sys.path:
    sys.path.insert(0, REF)
  from engine.engine import TorchEngine
  from models import construct_model
  rcfg = namedtuple('Config', CKPT_CFG.keys())(**dict(CKPT_CFG, resume=True))
  model, _ = construct_model(rcfg)
  eng = TorchEngine(model, rcfg, 'cpu', None, torch.load(ckpt_path, map_location='cpu', weights_only=False))
  assert eng.micro_steps == 4
  return [eng.step({'input_ids': tokens[i]}).item() for i in range(4, 8)]


def main():
  src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out')
  ck, cont = os.path.join(src, 'hip_ckpt_step_2.pth'), os.path.join(src, 'hip_ckpt_cont.npz')
  z = np.load(os.path.join(HERE, 'ckpt.npz'))
  tok = torch.from_numpy(z['tokens'])
  got = np.array(resume_with_reference(ck, tok))
  hip = np.load(cont)['losses']
  ref = z['losses'][4:]
  print('reference engine resumed from the HipEngine checkpoint:', np.round(got, 6))
  print('HipEngine continuing by itself                        :', np.round(hip, 6), ' rel', np.array2string(np.abs(got - hip) / hip, precision=2))
  print('reference engine resumed from ITS OWN checkpoint      :', np.round(ref, 6), ' rel', np.array2string(np.abs(got - ref) / ref, precision=2))
  assert (np.abs(got - hip) / hip).max() <= 1e-4 and (np.abs(got - ref) / ref).max() <= 1e-4
  if os.path.abspath(src) != HERE:
    shutil.copy(ck, os.path.join(HERE, 'hip_ckpt_step_2.pth'))
    shutil.copy(cont, os.path.join(HERE, 'hip_ckpt_cont.npz'))
    print('installed tests/golden/hip_ckpt_step_2.pth, hip_ckpt_cont.npz')


if __name__ == '__main__':
  main()
