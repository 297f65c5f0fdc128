import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
  config.addinivalue_line('markers', 'reference: needs /root/reference (build container only)')


def pytest_collection_modifyitems(config, items):
  """`-m gpu` tests SKIP where there is no GPU (the CPU-only build container) instead of failing in tapir_create."""
  try:
    import torch
    has_gpu = torch.cuda.is_available()
  except Exception:
    has_gpu = False
  if has_gpu:
    return
  skip = pytest.mark.skip(reason='needs a real MI355X (no GPU visible)')
  for item in items:
    if 'gpu' in item.keywords:
      item.add_marker(skip)
