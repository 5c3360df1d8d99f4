"""The instrumented kernel copies used by the timing experiments (tools/*_probe.sh, tools/ab_tail_builds.sh) are GENERATED from the production sources by committed
patches (tools/probes/<name>_probe.patch, tools/probes/gen_probe.sh) -- VERDICT r5 weak #11: hand-maintained copies drift, and a drifted probe silently
invalidates phase-removal evidence.  Checked here, on the CPU (cross-compile, no GPU):
  * every committed patch still APPLIES to today's production source;
  * the generated probe, built WITHOUT any -DV2X_*_DBG_BUILD flag, compiles to the production object's instruction stream, kernel by kernel (comments,
    labels and directives stripped) -- the instrumentation is provably inert when switched off, and nothing but instrumentation is in the patch."""
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "v2x-sim_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
PATCHES = sorted(os.path.basename(p)[:-len("_probe.patch")] for p in glob.glob(os.path.join(ROOT, "tools", "probes", "*_probe.patch")))


def _stream(src, out):
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-I", CSRC, src, "-o", out],
                          stderr=subprocess.DEVNULL)
    kernels = {}
    for m in re.finditer(r"^(_Z\w+):.*?s_endpgm", open(out).read(), flags=re.S | re.M):
        body = [re.sub(r"\s*;.*$", "", ln).strip() for ln in m.group(0).splitlines()[1:]]
        kernels[m.group(1)] = [ln for ln in body if ln and not ln.startswith((".", ";")) and not ln.endswith(":")]
    return kernels


def test_there_are_probes_and_no_hand_maintained_copies():
    assert PATCHES == ["conv_stream", "conv_tail"], PATCHES
    tracked = subprocess.run(["git", "ls-files", "tools/probes"], cwd=ROOT, capture_output=True, text=True).stdout.split()
    assert not [f for f in tracked if f.endswith(".hip")], "tools/probes/*.hip are generated files (gitignored), not sources"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("name", PATCHES)
def test_uninstrumented_probe_equals_production(name, tmp_path):
    gen = str(tmp_path / (name + "_probe.hip"))
    p = subprocess.run([os.path.join(ROOT, "tools", "probes", "gen_probe.sh"), name, gen], capture_output=True, text=True)
    assert p.returncode == 0, "the patch no longer applies to %s.hip: regenerate it (diff -u csrc/%s.hip <edited probe>)\n%s" % (name, name, p.stderr)
    assert "DBG_BUILD" in open(gen).read()          # it IS the instrumented copy
    prod = _stream(os.path.join(CSRC, name + ".hip"), str(tmp_path / "prod.s"))
    probe = _stream(gen, str(tmp_path / "probe.s"))
    assert set(prod) == set(probe) and prod, (sorted(set(prod) ^ set(probe)))
    for k in prod:
        assert prod[k] == probe[k], "%s: the un-instrumented probe's instruction stream differs from the production kernel's (%d vs %d instructions)" % (
            k, len(probe[k]), len(prod[k]))
