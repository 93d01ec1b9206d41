"""Input pipeline of the sharded evaluation (rpeflow_amd/loader.py; reference: DataLoader(num_workers=8) + copy_to_device,
eval_withocc.py:25-29, 56): batch order and contents must not depend on workers, threads vs processes, or the device ring."""
import threading
import time

import numpy as np
import pytest
import torch

from rpeflow_amd import evaluate as E
from rpeflow_amd.loader import InputPipeline
from rpeflow_amd.synthetic import SyntheticPairs


class Tiny(torch.utils.data.Dataset):
    """Sample i is recognisable from its contents; loading takes a (seeded) random while so workers finish out of order."""

    def __init__(self, n, delay=0.0, fail_at=None):
        self.n, self.delay, self.fail_at = n, delay, fail_at

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if self.delay:
            time.sleep(self.delay * ((i * 7919) % 5) / 4)
        if i == self.fail_at:
            raise ValueError("sample %d is broken" % i)
        return {"a": torch.full((3, 5), float(i)), "b": torch.tensor([i, 2 * i], dtype=torch.int64), "c": torch.full((4,), i % 251, dtype=torch.uint8)}


def check(batches, indices, batch_size):
    want = [indices[s:s + batch_size] for s in range(0, len(indices), batch_size)]
    assert len(batches) == len(want)
    for got, ids in zip(batches, want):
        assert got["a"].shape == (len(ids), 3, 5) and got["b"].dtype == torch.int64 and got["c"].dtype == torch.uint8
        assert got["a"][:, 0, 0].tolist() == [float(i) for i in ids]
        assert got["b"][:, 1].tolist() == [2 * i for i in ids]


@pytest.mark.parametrize("workers,processes", [(1, False), (4, False), (2, True)])
def test_order_and_contents_do_not_depend_on_the_workers(workers, processes):
    indices = list(range(1, 40, 3))  # 13 samples: a short last batch
    pipe = InputPipeline(Tiny(64, delay=0.004), indices, 4, "cpu", workers=workers, processes=processes)
    got = [{k: v.clone() for k, v in b.items()} for b in pipe]
    check(got, indices, 4)
    assert pipe.stats["batches"] == 4


def test_a_dataset_may_load_straight_into_the_host_batch():
    """``load_into(i, out)``: the pipeline hands the dataset the sample's place in the (pinned) host batch and copies nothing
    itself; batches equal the __getitem__ path's, whatever the number of workers."""
    class Direct(Tiny):
        def __init__(self, n):
            super().__init__(n)
            self.direct, self.lock = [], threading.Lock()

        def load_into(self, i, out):
            assert set(out) == {"a", "b", "c"} and out["a"].shape == (3, 5) and out["b"].dtype == torch.int64
            out["a"].fill_(float(i))
            out["b"].copy_(torch.tensor([i, 2 * i]))
            out["c"].fill_(i % 251)
            with self.lock:
                self.direct.append(i)

    indices = list(range(2, 50, 3))
    for workers in (1, 3):
        data = Direct(64)
        got = [{k: v.clone() for k, v in b.items()} for b in InputPipeline(data, indices, 4, "cpu", workers=workers)]
        check(got, indices, 4)
        assert sorted(data.direct) == indices[1:]  # (the first sample was read through __getitem__ to learn keys and shapes)


def test_pairs_hands_out_the_next_batch_early_and_recycles_slots():
    indices = list(range(30))
    pipe = InputPipeline(Tiny(30), indices, 4, "cpu", workers=3)
    seen, previous_next = [], None
    for batch, upcoming in pipe.pairs():
        if previous_next is not None:  # what was announced is what arrives, the very same tensors
            assert batch["a"] is previous_next["a"]
        seen.append({k: v.clone() for k, v in batch.items()})
        previous_next = upcoming
    assert previous_next is None
    check(seen, indices, 4)  # 8 batches through a ring of depth + workers = 6 host slots: slots were reused


def test_a_failing_sample_surfaces_in_the_consumer():
    pipe = InputPipeline(Tiny(20, fail_at=9), list(range(20)), 4, "cpu", workers=2)
    with pytest.raises(RuntimeError) as info:
        list(pipe)
    assert isinstance(info.value.__cause__, ValueError)


def test_leaving_the_loop_early_stops_the_threads():
    before = threading.active_count()
    pipe = InputPipeline(Tiny(400, delay=0.001), list(range(400)), 4, "cpu", workers=4)
    for n, _ in enumerate(pipe):
        if n == 2:
            break
    assert threading.active_count() == before


def test_empty_shard():
    assert list(InputPipeline(Tiny(4), [], 4, "cpu")) == []


def test_cached_synthetic_set_repeats_its_distinct_samples():
    data = SyntheticPairs(7, H=24, W=40, N=512, distinct=3, cache=True)
    assert data.prepare(threads=2) >= 0 and len(data.cache) == 3
    assert data[0]["pcs"] is data[3]["pcs"] and torch.equal(data[1]["images"], data[4]["images"])
    fresh = SyntheticPairs(7, H=24, W=40, N=512)
    assert torch.equal(data[5]["event_voxel"], fresh[2]["event_voxel"])  # 5 % 3 == 2: the generator's sample 2


def test_evaluate_is_independent_of_the_pipeline():
    from tests.test_evaluate import fake_model
    data = SyntheticPairs(9, H=24, W=40, N=512)
    ref = E.new_accumulator("cpu")
    for s in range(0, 9, 2):
        batch = E.collate([data[i] for i in range(s, min(s + 2, 9))])
        E.accumulate(ref, fake_model(batch), batch)
    for workers, processes in ((1, False), (3, False), (2, True)):
        stats = {}
        _, acc = E.evaluate(fake_model, data, 2, "cpu", workers=workers, processes=processes, stats=stats)
        assert np.array_equal(acc.numpy(), ref.numpy())
        assert stats["batches"] == 5 and stats["shard"] == 9


@pytest.mark.gpu
@pytest.mark.parametrize("pin", [False, True])
def test_device_ring_delivers_the_samples(pin):
    """Pinned ring + copy stream (and the no-staging path for pinned samples): 12 batches through 3 device slots."""
    data = SyntheticPairs(23, H=64, W=96, N=1024, distinct=5, cache=True, pin=pin)
    pipe = InputPipeline(data, list(range(23)), 2, "cuda:0", workers=3)
    n = 0
    for batch, upcoming in pipe.pairs():
        ids = list(range(2 * n, min(2 * n + 2, 23)))
        for k in ("images", "event_voxel", "pcs", "flow_2d"):
            want = torch.stack([data[i][k] for i in ids])
            assert batch[k].device.type == "cuda" and torch.equal(batch[k].cpu(), want), (n, k)
        if upcoming is not None:
            assert torch.equal(upcoming["pcs"][0].cpu(), data[2 * n + 2]["pcs"])
        (batch["event_voxel"] * 2).sum()  # consumer work on the compute stream before the slot goes back
        n += 1
    assert n == 12 and pipe.stats["batches"] == 12 and (pipe.stats["direct"] == 12) == pin


@pytest.mark.gpu
def test_load_into_fills_the_pinned_ring_on_a_gpu_run():
    """``load_into`` on a CUDA run: the places handed to the dataset are PINNED host memory (the H2D copy starts from them),
    every batch arrives on the device with the __getitem__ path's contents, through more batches than the ring has slots."""
    pinned_places = []  # (list.append is atomic: three loader threads call load_into)

    class Direct(SyntheticPairs):
        def load_into(self, i, out):
            sample = self[i]
            assert set(out) == set(sample)
            pinned_places.append(all(v.is_pinned() for v in out.values()))
            for k, v in sample.items():
                out[k].copy_(v)

    data = Direct(23, H=64, W=96, N=1024, distinct=5, cache=True)
    pipe = InputPipeline(data, list(range(23)), 2, "cuda:0", workers=3)
    n = 0
    for batch, _ in pipe.pairs():
        ids = list(range(2 * n, min(2 * n + 2, 23)))
        for k in ("images", "event_voxel", "pcs", "flow_2d"):
            assert batch[k].device.type == "cuda" and torch.equal(batch[k].cpu(), torch.stack([data[i][k] for i in ids])), (n, k)
        n += 1
    assert n == 12 and len(pinned_places) == 22 and all(pinned_places)  # every sample but the first, which defined keys and shapes through __getitem__


def test_raw_events_need_the_device_stage():
    """A dataset that returns raw events (flyingthings3d.py:206-208 without a pre-processed file) is voxelised by the pipeline's
    copy stage on the GPU; there is no CPU voxelisation in the product, so a CPU consumer is refused -- loudly."""
    data = SyntheticPairs(4, H=32, W=48, N=64, events=2000)
    assert "events" in data[0] and "event_voxel" not in data[0] and data[0]["events"].dtype == torch.float32
    assert data[0]["events"].shape[0] != data[1]["events"].shape[0] <= 2000  # ragged
    with pytest.raises(RuntimeError, match="voxelised on the GPU"):
        list(InputPipeline(data, [0, 1, 2, 3], 2, "cpu", workers=1))


@pytest.mark.gpu
@pytest.mark.parametrize("pin", [False, True])
def test_raw_events_are_voxelised_on_the_device_bit_for_bit(pin):
    """events [n,4] float32 per sample (ragged) -> pinned ring -> device -> event_ops.events_to_voxel on the copy stream: the
    consumer's event_voxel equals eventsToVoxel(events, 10 bins, polarity split) -- the CPU restatement pinned to the reference's
    float32 goldens -- for every sample, short last batch included, with and without the staging ring."""
    from oracle import oracle as O
    H, W = 40, 56
    data = SyntheticPairs(11, H=H, W=W, N=256, cache=True, pin=pin, events=6000)
    pipe = InputPipeline(data, list(range(11)), 3, "cuda:0", workers=2)
    n = 0
    for batch, upcoming in pipe.pairs():
        ids = list(range(3 * n, min(3 * n + 3, 11)))
        assert "events" not in batch and batch["event_voxel"].shape == (len(ids), 20, H, W) and batch["event_voxel"].device.type == "cuda"
        got = batch["event_voxel"].cpu().numpy()
        for s, i in enumerate(ids):
            want = O.events_to_voxel(data[i]["events"].numpy(), 10, H, W, True)
            assert np.array_equal(got[s].view(np.uint32), want.view(np.uint32)), (n, s)
            assert torch.equal(batch["pcs"][s].cpu(), data[i]["pcs"])
        n += 1
    assert n == 4 and (pipe.stats["direct"] == 4) == pin


@pytest.mark.gpu
def test_raw_events_outside_the_sensor_surface_in_the_consumer():
    class Bad(SyntheticPairs):
        def __getitem__(self, i):
            s = dict(super().__getitem__(i))
            if i == 2:
                s["events"] = s["events"].clone()
                s["events"][5, 0] = 9999.0
            return s
    with pytest.raises(RuntimeError) as err:
        list(InputPipeline(Bad(4, H=32, W=48, N=64, events=500), [0, 1, 2, 3], 2, "cuda:0", workers=1))
    assert isinstance(err.value.__cause__, IndexError)
