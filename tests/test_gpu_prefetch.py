"""GPU: datasets.DevicePrefetcher (row f-2) -- batches arrive in order, bit-equal, on the device; errors of the producer
surface in the consumer; logits from host-resident sweeps (copied inline or prefetched on their own stream) equal the HBM-resident ones."""
import importlib.util
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_prefetcher_order_values_and_errors(device):
    from v2x_sim_amd.datasets import DevicePrefetcher
    rng = np.random.default_rng(0)
    batches = [{"points": rng.standard_normal((3, 1000, 4)).astype(np.float32), "n": torch.tensor([i, i + 1]),
                "meta": ("frame", i), "nested": [torch.full((5,), float(i))]} for i in range(7)]
    got = []
    for b in DevicePrefetcher(iter(batches), device, depth=2):
        assert b["points"].device.type == "cuda" and b["n"].device.type == "cuda" and b["meta"][0] == "frame"
        got.append((b["points"].cpu().numpy(), int(b["n"][0]), b["meta"][1], float(b["nested"][0][0])))
    assert len(got) == 7
    for i, (p, n, m, f) in enumerate(got):
        assert np.array_equal(p, batches[i]["points"]) and n == i and m == i and f == float(i)

    def bad():
        yield {"x": np.zeros(3, np.float32)}
        raise ValueError("reader broke")
    with pytest.raises(ValueError, match="reader broke"):
        for _ in DevicePrefetcher(bad(), device):
            pass
    with pytest.raises(RuntimeError):
        DevicePrefetcher(iter([]), "cpu")


def test_streaming_from_host_gives_the_resident_logits(device):
    """The three input arms of tools/stream_points.py (sweeps resident in HBM, copied inside the step, prefetched on a copy stream by
    DevicePrefetcher) must produce the same logits bit for bit: a copy overlapped with the previous step's kernels may not be read early.
    (The RATES of the three arms are a measurement, not a parity property: bench.py prints them as `host_streaming`.)"""
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    spec = importlib.util.spec_from_file_location("stream_points", os.path.join(tools, "stream_points.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res, outs = mod.main(8, 6, keep_outputs=True)
    assert set(res) == {"resident", "inline", "prefetch"} and all(v > 0 for v in res.values())
    # a loose STRUCTURAL bound, not a performance claim: a regression that serialises the copy stream behind the compute stream -- or
    # blocks the producer thread -- would show as the prefetched arm falling far below the resident one (each arm is the best of two passes;
    # measured 0.85-0.95 at this size).  The gated number stays bench.py's `host_streaming`.
    assert res["prefetch"] >= 0.4 * res["resident"], res
    for arm in ("inline", "prefetch"):
        for k in ("cls", "loc"):
            assert torch.isfinite(outs[arm][k]).all() and torch.equal(outs[arm][k], outs["resident_same_ring_slot"][k]), (arm, k)
