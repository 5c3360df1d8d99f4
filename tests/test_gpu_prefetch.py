"""GPU: datasets.DevicePrefetcher (row f-2) -- batches arrive in order, bit-equal, on the device; errors of the producer
surface in the consumer; with the copies on their own stream the points -> logits rate from host memory stays close to the
HBM-resident rate."""
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


def test_streaming_from_host_keeps_up(device):
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    spec = importlib.util.spec_from_file_location("stream_points", os.path.join(tools, "stream_points.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # a throughput comparison on a shared host: 8 steps take ~0.1 s, one scheduling hiccup of the producer thread is 10 % -- best of three
    tries = []
    for _ in range(3):
        res, out = mod.main(32, 8)
        assert torch.isfinite(out["cls"]).all()
        tries.append(res)
        if res["prefetch"] >= 0.8 * res["resident"] and res["prefetch"] >= res["inline"] * 0.98:
            break
    else:
        raise AssertionError("the prefetched rate stayed below 80 %% of the resident rate / below the inline-copy rate in three runs: %s" % tries)
