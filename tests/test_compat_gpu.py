"""-m gpu: the reference's call sequence on the import shims (diffusers / pynvml / cv2 stand-ins) with its exact FSDP
wrap arguments (`fsdp_chunked_coherent.py:63-88`) around the parameter-less HIP modules — one process, world-1 RCCL."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_call_sequence_runs_on_the_shims_with_fsdp_wrap(gpu, tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = str(tmp_path / "out.mp4")
    r = subprocess.run([sys.executable, "-m", "vdx.compat.run", os.path.join(ROOT, "tests", "compat_reference_style.py"), out],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("COMPAT-OK")]
    assert line, r.stdout[-2000:]
    assert "unet FullyShardedDataParallel" in line[0] and "in_channels 4" in line[0] and "frames 5" in line[0]
    assert os.path.getsize(out) > 1000
