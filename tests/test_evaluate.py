"""Eval accumulators (eval_withocc.py:65-135, eval_noocc.py:57-116) and the sharded harness:
CPU tests, including the 2-process gloo run of the one collective."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from rpeflow_amd import evaluate as E
from rpeflow_amd.synthetic import SyntheticPairs, frame_pair

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_style(outputs, inputs):
    """The reference's per-sample loop, restated with numpy and Python floats (eval_withocc.py:65-108)."""
    m2 = dict(c=0, e=0.0, a=0.0, f=0.0); m3 = dict(c=0, e=0.0, a=0.0, b=0.0); mn = dict(c=0, e=0.0, a=0.0, b=0.0)
    for i in range(outputs["flow_2d"].shape[0]):
        p2, p3 = outputs["flow_2d"][i].numpy(), outputs["flow_3d"][i].numpy()
        t2, t3 = inputs["flow_2d"][i].numpy(), inputs["flow_3d"][i].numpy()
        k2 = t2[2] > 0 if t2.shape[0] > 2 else np.ones(t2.shape[1:], bool)
        k3 = t3[3] > 0 if t3.shape[0] > 3 else np.ones(t3.shape[1], bool)
        t2, t3 = t2[:2], t3[:3]
        e2 = np.sqrt(((p2 - t2) ** 2).sum(0)); e3 = np.sqrt(((p3 - t3) ** 2).sum(0))
        k2 &= ~np.isnan(e2); k3 &= ~np.isnan(e3)
        with np.errstate(divide="ignore", invalid="ignore"):
            fl = (e2 > 3.0) & (e2 / np.linalg.norm(t2, axis=0) > 0.05)
        m2["c"] += int(k2.sum()); m2["e"] += float(e2[k2].sum()); m2["a"] += int((e2[k2] < 1.0).sum()); m2["f"] += float(fl[k2].sum())
        m3["c"] += int(k3.sum()); m3["e"] += float(e3[k3].sum()); m3["a"] += int((e3[k3] < 0.05).sum()); m3["b"] += int((e3[k3] < 0.1).sum())
        if "occ_mask_3d" in inputs:
            kn = k3 & (inputs["occ_mask_3d"][i].numpy() == 0)
            mn["c"] += int(kn.sum()); mn["e"] += float(e3[kn].sum()); mn["a"] += int((e3[kn] < 0.05).sum()); mn["b"] += int((e3[kn] < 0.1).sum())
    return np.array([m2["c"], m2["e"], m2["a"], m2["f"], m3["c"], m3["e"], m3["a"], m3["b"], mn["c"], mn["e"], mn["a"], mn["b"]], np.float64)


def fake_model(batch):
    """Deterministic stand-in: the targets plus a sample-dependent error (and a few NaNs to exercise the mask)."""
    f2 = batch["flow_2d"][:, :2].float() + 0.8 * torch.sin(batch["event_voxel"][:, :2].float() * 3.0)
    f2 = f2 + 6.0 * (batch["event_voxel"][:, 2:4].float() > 1.2).float()  # a tenth of the pixels far off: the Fl criterion fires
    f3 = batch["flow_3d"][:, :3].float() + 0.06 * torch.cos(batch["pcs"][:, :3].float() * 5.0)
    f3[:, :, ::97] = float("nan")
    return {"flow_2d": f2, "flow_3d": f3}


@pytest.mark.parametrize("dsec", [False, True])
def test_accumulators_match_reference_style_loop(dsec):
    data = SyntheticPairs(3, H=24, W=40, N=512, dsec=dsec)
    batch = E.collate([data[i] for i in range(3)])
    out = fake_model(batch)
    acc = E.accumulate(E.new_accumulator("cpu"), out, batch)
    ref = reference_style(out, batch)
    counts = [0, 2, 3, 4, 6, 7, 8, 10, 11]
    assert np.array_equal(acc.numpy()[counts], ref[counts])  # counts and threshold hits: exact
    np.testing.assert_allclose(acc.numpy()[[1, 5, 9]], ref[[1, 5, 9]], rtol=1e-6)  # fp32 maps, float64 sums
    m = E.finalize(acc)
    assert ("EPE3D_noc" in m) == (not dsec)
    assert abs(m["EPE2D"] - ref[1] / ref[0]) < 1e-6


def test_known_answer():
    inputs = {"flow_2d": torch.tensor([[[[3.0, 0.0]], [[4.0, 0.0]], [[1.0, 0.0]]]]),  # [1,3,1,2]: second pixel masked out
              "flow_3d": torch.zeros(1, 3, 4), "occ_mask_3d": torch.tensor([[0.0, 1.0, 0.0, 0.0]])}
    outputs = {"flow_2d": torch.zeros(1, 2, 1, 2), "flow_3d": torch.tensor([[[0.03, 0.0, 0.2, 0.0], [0.0, 0.08, 0.0, 0.0], [0.0, 0.0, 0.0, float("nan")]]])}
    a = E.accumulate(E.new_accumulator("cpu"), outputs, inputs).tolist()
    assert a[:4] == [1.0, 5.0, 0.0, 1.0]                       # one valid pixel, EPE 5, not <1px, Fl hit
    assert a[4] == 3.0 and abs(a[5] - 0.31) < 1e-6 and a[6:8] == [1.0, 2.0]   # NaN point dropped
    assert a[8] == 2.0 and abs(a[9] - 0.23) < 1e-6 and a[10:12] == [1.0, 1.0]  # occluded point dropped too


def test_shards_partition_the_dataset_without_padding():
    for n in (0, 1, 5, 8, 13):
        for w in (1, 2, 3, 8):
            parts = [E.shard_indices(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))


def test_single_process_evaluate():
    data = SyntheticPairs(5, H=24, W=40, N=512)
    metrics, acc = E.evaluate(fake_model, data, batch_size=2, device="cpu")
    whole = E.collate([data[i] for i in range(5)])
    ref = reference_style(fake_model(whole), whole)
    np.testing.assert_allclose(acc.numpy(), ref, rtol=1e-6)
    assert metrics["counts"]["2d"] == 5 * 24 * 40


WORKER = r'''
import os, sys, json, torch
sys.path.insert(0, os.environ["RPE_ROOT"])
import torch.distributed as dist
from rpeflow_amd import evaluate as E
from rpeflow_amd.synthetic import SyntheticPairs
from tests.test_evaluate import fake_model
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
data = SyntheticPairs(5, H=24, W=40, N=512)            # 5 samples over 2 ranks: shards of 3 and 2
metrics, acc = E.evaluate(fake_model, data, batch_size=2, device="cpu", rank=rank, world_size=world)
print("RESULT", rank, json.dumps(acc.tolist()))
if rank == 0:  # an unsharded call inside the job (rank 0 validating on its own): no collective, nothing multiplied, no waiting for rank 1
    _, alone = E.evaluate(fake_model, data, batch_size=2, device="cpu")
    print("ALONE", rank, json.dumps(alone.tolist()))
# the first forward in turn (MIOpen's solver search on a GPU run): rank 0's step ends before any other rank's begins
import time
from rpeflow_amd.evaluate import first_forward_in_turn
def step():
    t0 = time.time()
    time.sleep(0.3)
    return t0, time.time()
span = first_forward_in_turn(step, rank, dist.group.WORLD)
print("TURN", rank, json.dumps(span))
assert first_forward_in_turn(lambda: 7, rank, None) == 7   # no group: just the call
dist.barrier()
dist.destroy_process_group()
'''


def test_two_process_gloo_all_reduce_equals_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RPE_ROOT=ROOT,
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-500:] for o in outs]
    import json
    accs = [json.loads(next(l for l in o[0].splitlines() if l.startswith("RESULT")).split(" ", 2)[2]) for o in outs]
    assert accs[0] == accs[1]  # every rank holds the global sums
    data = SyntheticPairs(5, H=24, W=40, N=512)
    _, single = E.evaluate(fake_model, data, batch_size=2, device="cpu")
    np.testing.assert_allclose(np.array(accs[0]), single.numpy(), rtol=1e-12)
    assert accs[0][0] == single[0].item() and accs[0][4] == single[4].item()
    alone = json.loads(next(l for l in outs[0][0].splitlines() if l.startswith("ALONE")).split(" ", 2)[2])
    assert alone == single.tolist()  # world_size=1, group=None inside an initialised 2-rank group: the plain single-process sums
    turns = [json.loads(next(l for l in o[0].splitlines() if l.startswith("TURN")).split(" ", 2)[2]) for o in outs]
    assert turns[0][1] <= turns[1][0]  # first_forward_in_turn: rank 0 had finished its step when rank 1 began its own


@pytest.mark.parametrize("device", ["cpu", pytest.param("cuda:0", marks=pytest.mark.gpu)])
def test_accumulators_match_the_reference_loops(device):
    """tests/golden/eval_accumulators.npz: the metric sums the reference's own Evaluator.run loops (eval_withocc.py:45-135,
    eval_noocc.py:45-116, run unmodified by tests/golden/make_golden.py) reach on these frame pairs and predictions."""
    golden = np.load(os.path.join(ROOT, "tests", "golden", "eval_accumulators.npz"))
    for dsec, key, n in ((False, "withocc", 12), (True, "noocc", 8)):
        data = SyntheticPairs(5, H=24, W=40, N=512, dsec=dsec)
        acc = E.new_accumulator(device)
        for idx in ((0, 1), (2, 3), (4,)):
            batch = {k: v.to(device) for k, v in E.collate([data[i] for i in idx]).items()}
            E.accumulate(acc, fake_model(batch), batch)
        acc = acc.cpu()
        got, want = acc.numpy()[:n], golden[key]
        hits = [0, 2, 3, 4, 6, 7, 8, 10, 11][:6 if n == 8 else 9]
        assert np.array_equal(got[hits], want[hits]), (key, got, want)          # counts and threshold hits: exact
        sums = [i for i in range(n) if i not in hits]
        np.testing.assert_allclose(got[sums], want[sums], rtol=2e-6)             # fp32 per-sample sums there, float64 here
        assert want[3] > 0 and want[2] > 0                                       # the case exercises 1px and Fl
        if n == 8:
            assert acc.numpy()[8:].sum() == 0                                    # no occlusion group without occ_mask_3d


@pytest.mark.gpu
def test_graph_replayed_evaluation_matches_eager():
    """The harness's HIP-graph path (static input buffers, one graph per batch shape, short last batch included) gives
    the metrics of the eager loop."""
    from rpeflow_amd.model import RPEFlow
    torch.manual_seed(0)
    model = RPEFlow().to("cuda:0").eval()
    data = SyntheticPairs(5, H=128, W=192, N=8192)
    eager, acc_e = E.evaluate(model, data, batch_size=2, device="cuda:0", graph=False)
    graphed, acc_g = E.evaluate(model, data, batch_size=2, device="cuda:0", graph=True)
    assert eager["counts"] == graphed["counts"]
    a_g, a_e = acc_g.cpu().numpy(), acc_e.cpu().numpy()
    sums = [E.FIELDS.index(f) for f in E.FIELDS if f.startswith(("count", "epe"))]
    hits = [i for i in range(len(E.FIELDS)) if i not in sums]
    np.testing.assert_allclose(a_g[sums], a_e[sums], rtol=1e-4)
    # threshold hit counts: the library convs may pick another solver under capture (~1e-5 on the flow), which can
    # move a few points of a random-weight model across a threshold
    np.testing.assert_allclose(a_g[hits], a_e[hits], atol=16, rtol=1e-3)


@pytest.mark.gpu
def test_evaluation_from_raw_events_equals_evaluation_from_their_voxel_grids():
    """A set that hands out raw events (flyingthings3d.py:206-208 without a pre-processed file; voxelised by the pipeline's copy
    stage) against the same set with the grids made beforehand by event_ops.events_to_voxel: the same metrics, sum for sum."""
    from rpeflow_amd.event_ops import events_to_voxel
    from rpeflow_amd.model import RPEFlow

    class Voxelised(torch.utils.data.Dataset):
        def __init__(self, raw):
            self.raw, self.f = raw, raw.event_format

        def __len__(self):
            return len(self.raw)

        def __getitem__(self, i):
            s = dict(self.raw[i])
            ev = s.pop("events")
            s["event_voxel"] = events_to_voxel(ev.to("cuda:0"), self.f["bins"], self.f["height"], self.f["width"], self.f["polarity"]).cpu()
            return s

    torch.manual_seed(0)
    model = RPEFlow().to("cuda:0").eval()
    raw = SyntheticPairs(5, H=128, W=192, N=8192, events=40000, cache=True)
    from_events, acc_a = E.evaluate(model, raw, batch_size=2, device="cuda:0", graph=False)
    from_grids, acc_b = E.evaluate(model, Voxelised(raw), batch_size=2, device="cuda:0", graph=False, workers=1)
    assert from_events["counts"] == from_grids["counts"] and np.array_equal(acc_a.cpu().numpy(), acc_b.cpu().numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("dsec,B,H,W,N", [(False, 3, 24, 40, 517), (True, 2, 31, 33, 1000), (False, 4, 544, 960, 8192), (True, 3, 480, 640, 8192)])
def test_device_accumulate_kernel_equals_the_tensor_form(dsec, B, H, W, N):
    """rpe_eval_accumulate (csrc/eval.hip) against the host-side tensor statement of eval_withocc.py:65-108 on the same
    batch: counts and threshold hits exactly, EPE sums to float64 re-association; NaN predictions, masked targets,
    zero-length ground-truth vectors (epe / |gt| = inf) and occlusion masks included; twice into the same accumulator."""
    g = torch.Generator().manual_seed(B * 1000 + N)
    t2 = torch.randn(B, 2, H, W, generator=g) * 5.0
    t2[:, :, ::7, ::5] = 0.0                                                 # |gt| = 0
    t3 = torch.randn(B, 3, N, generator=g) * 0.2
    inputs = {"flow_2d": torch.cat([t2, (torch.rand(B, 1, H, W, generator=g) < 0.8).float()], 1), "flow_3d": t3}
    if dsec:
        inputs["flow_3d"] = torch.cat([t3, (torch.rand(B, 1, N, generator=g) < 0.9).float()], 1)
    else:
        inputs["occ_mask_3d"] = (torch.rand(B, N, generator=g) < 0.2).float()
    out = {"flow_2d": t2 + torch.randn(B, 2, H, W, generator=g) * torch.tensor([0.3, 4.0]).view(1, 2, 1, 1),
           "flow_3d": t3 + torch.randn(B, 3, N, generator=g) * 0.05}
    out["flow_2d"][:, 0, 3::11, 2::13] = float("nan")
    out["flow_3d"][:, 2, ::97] = float("nan")
    ref = E.new_accumulator("cpu")
    E.accumulate(ref, out, inputs)
    E.accumulate(ref, out, inputs)
    acc = E.new_accumulator("cuda:0")
    d = lambda t: {k: v.to("cuda:0") for k, v in t.items()}
    E.accumulate(acc, d(out), d(inputs))
    E.accumulate(acc, {k: v.to("cuda:0").transpose(0, 1).contiguous().transpose(0, 1) for k, v in out.items()}, d(inputs))  # strided predictions
    got, want = acc.cpu().numpy(), ref.numpy()
    hits = [0, 2, 3, 4, 6, 7, 8, 10, 11]
    assert np.array_equal(got[hits], want[hits]), (got, want)
    # the kernel's sqrt is correctly rounded (as torch's is on a GPU tensor, where the reference evaluates); the CPU tensor op
    # is a vectorised approximation an ulp off on some inputs, so the EPE sums agree to fp32 ulps, not float64 ones
    np.testing.assert_allclose(got[[1, 5, 9]], want[[1, 5, 9]], rtol=2e-7)
    assert want[3] > 0 and want[2] > 0 and (dsec or want[8] > 0)
