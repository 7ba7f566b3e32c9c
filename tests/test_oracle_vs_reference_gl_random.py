"""Out-of-sample differential test: the CPU oracle against the REFERENCE ITSELF, on states no fixture holds.

tests/golden/gl_*.npz pin the oracle on 101 committed states; this test draws fresh (family, seed, steps, domain_rand, view)
triples from a seeded generator of its own, runs /root/reference/miniworld unmodified on Mesa llvmpipe
(tools/refshim_gl.py) and demands oracle == reference bit for bit: RGB, the 16-bit depth buffer and the float32 depth map.
Every registered family (/root/reference/miniworld/envs/__init__.py:44-157) appears at least once.  It is the check a
reviewer does by hand with tools/debug/gl_vs_oracle.py.

Needs /root/reference and the Mesa software driver: it runs in the build container and is skipped on the GPU box (which has
neither).  The comparison runs in a process of its own: the GL shim and the stub-GL shim of the other CPU tests cannot share one.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "ref_random_diff.py")
HAVE_REF = os.path.isdir("/root/reference/miniworld") and os.path.exists("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so")
pytestmark = pytest.mark.skipif(not HAVE_REF, reason="needs /root/reference and Mesa's swrast_dri.so (the build container)")


def _run(*args):
    r = subprocess.run([sys.executable, TOOL, *args], capture_output=True, text=True, timeout=1500)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, f"the worker printed no result (rc {r.returncode}):\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}"
    return r.returncode, json.loads(lines[-1])


def test_oracle_equals_the_reference_on_unseen_states():
    """46 triples (every one of the 23 families twice), 4 samples per pixel: llvmpipe's GL_MAX_SAMPLES, what the reference's
    FrameBuffer falls back to (opengl.py:229-231)."""
    rc, j = _run("--cases", "46", "--rng-seed", "60601")
    assert "skipped" not in j
    assert "llvmpipe" in j["driver"] and j["samples"] == 4
    assert j["cases"] == 46 and len(j["families"]) == 23
    assert any(r["top"] for r in j["results"]) and any(r["domain_rand"] for r in j["results"]) and any(r["steps"] > 30 for r in j["results"])
    assert all(1.0 < r["mean"] < 254.0 for r in j["results"])            # real frames
    assert not j["bad"] and rc == 0, f"oracle != reference on {[(b['cls'], b['seed'], b['steps'], b['domain_rand'], b['top'], b['rgb_bad'], b['z_bad'], b['depth_bad']) for b in j['bad']]}"


def test_oracle_equals_the_reference_single_sampled_fallback_on_unseen_states():
    """The other fallback (opengl.py:263-284): glTexImage2DMultisample fails, FrameBuffer renders single-sampled."""
    rc, j = _run("--cases", "4", "--rng-seed", "60602", "--one-spp")
    assert "skipped" not in j and j["samples"] == 1 and j["cases"] == 4
    assert not j["bad"] and rc == 0, f"oracle != reference (1 sample) on {[(b['cls'], b['seed'], b['steps'], b['rgb_bad'], b['z_bad']) for b in j['bad']]}"
