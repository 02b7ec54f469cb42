"""GPU: the opt-in duo ring GEMM (csrc/gemm_duo.h; OCRVI_GEMM_DUO=1, read once per process -> child process).  Matches the Linears of
model/rec2/svtrv2.py:28-39,77-86 and the 1x1 convolutions of torchvision's Bottleneck as called from model/det/backbone.py:34-37, like
the ring GEMM whose shapes it takes."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_duo_ring_gemm_epilogues_in_a_child_process():
    env = dict(os.environ)
    env["OCRVI_GEMM_DUO"] = "1"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_duo_cases.py")], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count(" ok") >= 9
