#!/usr/bin/env python3
"""|dEPE| of the BENCHED configuration (bench.py: batch 4 of 544x960 + 8192 points, device IDS, forward_ahead in one HIP
graph, seeded parameters) against the reference's CPU output (tests/golden/model_bench_b4_544x960.npz), under whatever
MIOpen solver environment the process was started with; also ms per replay.  One JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from rpeflow_amd.model import RPEFlow  # noqa: E402
from rpeflow_amd.synthetic import load_seeded_parameters  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(ROOT, "tests", "golden", "model_bench_b4_544x960.npz"))
    model = load_seeded_parameters(RPEFlow()).to(dev).eval()
    batch = bench.make_batch(4, dev, first_seed=1000)
    res = {"env": {k: os.environ.get(k) for k in ("MIOPEN_DEBUG_CONV_WINOGRAD", "MIOPEN_DEBUG_CONV_FFT", "MIOPEN_DEBUG_CONV_DIRECT",
                                                   "MIOPEN_DEBUG_CONV_GEMM", "MIOPEN_DEBUG_CONV_IMPLICIT_GEMM")}}

    def deltas(out):
        return bench.golden_epe_delta(out, batch, g)

    for _ in range(2):
        out = model(batch)
    torch.cuda.synchronize()
    res["eager"] = deltas(out)
    ids = torch.cat(model._clouds(batch, *model._cameras(batch)), 0).cpu().numpy()
    gold = np.concatenate([g["pc1_ids"], g["pc2_ids"]], 0)
    res["ids_coordinates_differing"] = int((ids.view(np.uint32) != gold.view(np.uint32)).sum())
    order = model.sample_order(batch)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        out = model.forward_ahead(batch, order, batch)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    res["graph"] = deltas(out)
    t0 = time.perf_counter()
    for _ in range(20):
        graph.replay()
    torch.cuda.synchronize()
    res["ms_per_step"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
