"""Generate tests/golden/*.npz by running the REFERENCE's own CPU/PyTorch path.

Runs only in the build container, where /root/reference is mounted:

    python -m tests.golden.make_golden            (from the repo root)

It imports ``models`` from /root/reference (nothing is copied), feeds it the
seeded inputs of tests/cases.py and stores the outputs.  The .npz files are
data: reference outputs (and nothing of the reference's source).  The GPU box
has no /root/reference; tests there read the committed .npz only.
"""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
REF = os.environ.get("RPEFLOW_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import torch  # noqa: E402

import contextlib, io  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):  # "Failed to load CUDA extensions" notice
    from models import csrc as ref_ops  # noqa: E402
    from models import utils as ref_utils  # noqa: E402
    from models import pointconv as ref_pc  # noqa: E402
    from models import pwc3d_core as ref_3d  # noqa: E402

from tests import cases as K  # noqa: E402
from tests import inputs as I  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
T = torch.from_numpy


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def gen_knn():
    for name in K.KNN_CASES:
        inp, qry, k = K.knn_inputs(name)
        d = ref_ops.squared_distance(T(qry), T(inp))
        vals, idx = d.topk(k, dim=2, largest=False)
        idx2 = ref_ops.k_nearest_neighbor(T(inp), T(qry), k)
        assert torch.equal(idx, idx2)
        # the (k+1)-th smallest distance tells the checker whether the k|k+1 boundary is a tie
        kk = min(k + 1, inp.shape[1])
        nxt = d.topk(kk, dim=2, largest=False).values[..., -1]
        save(name, idx=idx.numpy().astype(np.int32), dist=vals.numpy(), next_dist=nxt.numpy())


def gen_fps():
    for name in K.FPS_CASES:
        xyz, S = K.fps_inputs(name)
        idx = ref_ops.furthest_point_sampling(T(xyz), S)
        save(name, idx=idx.numpy().astype(np.int32))


def gen_sqdist():
    for name in K.SQDIST_CASES:
        a, b = K.sqdist_inputs(name)
        save(name, dist=ref_ops.squared_distance(T(a), T(b)).numpy())


def gen_corr():
    for name in K.CORR_CASES:
        a, b, md = K.corr_inputs(name)
        save(name, out=ref_ops.correlation2d(T(a), T(b), md).numpy())
        # gradients of the reference's differentiable fallback (_correlation_py through autograd)
        ta, tb = T(a).clone().requires_grad_(True), T(b).clone().requires_grad_(True)
        ref_ops.correlation2d(ta, tb, md).backward(T(K.corr_grad_output(name)))
        save(name + "_grad", grad1=ta.grad.numpy(), grad2=tb.grad.numpy())


def gen_glue():
    d = {k: T(v) for k, v in K.glue_inputs().items()}
    out = {}
    out["gather_cf"] = ref_utils.batch_indexing_channel_first(d["feat_3d"], d["idx"])
    out["gather_cl"] = ref_utils.batch_indexing_channel_last(d["feat_3d"].transpose(1, 2).contiguous(), d["idx"])
    out["backwarp_2d"] = ref_utils.backwarp_2d(d["feat_2d"], d["flow"], padding_mode="border")
    out["grid_sample_wrapper"] = ref_utils.grid_sample_wrapper(d["feat_2d"], d["xy"])
    out["knn_interp"] = ref_utils.knn_interpolation(d["xyz"], d["feat_3d"], d["xyz_q"], k=3)
    out["backwarp_3d"] = ref_utils.backwarp_3d(d["xyz"], d["xyz"] + 0.1, d["flow3"], k=3)
    ref_utils.mesh_grid_cache.clear()
    out["project_feat"] = ref_utils.project_feat_with_nn_corr(d["xy"], d["feat_2d"], d["feat_3d"])
    save("glue_ops", **{k: v.numpy() for k, v in out.items()})


def gen_events():
    # event_utils.py imports h5py and cv2 at module level (HDF5 loading, visualisation: :11-20, :306-440); neither is
    # installed here and neither is touched by eventsToVoxel, which is numpy + torch only.  Empty placeholder modules
    # satisfy the import statements; the function under test runs unmodified.
    import types
    if "h5py" not in sys.modules:
        sys.modules["h5py"] = types.ModuleType("h5py")
    if "cv2" not in sys.modules:
        cv2 = types.ModuleType("cv2")
        cv2.setNumThreads = lambda n: None
        cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda flag: None)
        sys.modules["cv2"] = cv2
    with contextlib.redirect_stdout(io.StringIO()):
        import event_utils as ref_events
    for name in K.EVENT_CASES:
        ev, H, W, bins, pol = K.event_inputs(name)
        vox = ref_events.eventsToVoxel(ev.copy(), num_bins=bins, height=H, width=W, event_polarity=pol, temporal_bilinear=True)
        save(name, voxel=np.asarray(vox, np.float32))


def load_params(module, seed):
    shapes = [(k, tuple(v.shape)) for k, v in module.state_dict().items()]
    params = I.fill_params(shapes, seed)
    module.load_state_dict({k: T(v) for k, v in params.items()}, strict=True)
    module.eval()
    return params


@torch.no_grad()
def gen_blocks():
    c = K.BLOCK_CASES["pointconv_down"]
    m = ref_pc.PointConvDownSampling(c["C"], c["Cout"], norm=c["norm"], k=c["k"])
    load_params(m, c["seed"] + 1000)
    x = K.block_inputs("pointconv_down")
    save("pointconv_down", out=m(T(x["xyz"]), T(x["feat"]), T(x["sampled"])).numpy())

    c = K.BLOCK_CASES["pointconv_nosample"]
    m = ref_pc.PointConvNoSampling(c["C"], c["Cout"], norm=c["norm"], k=c["k"])
    load_params(m, c["seed"] + 1000)
    x = K.block_inputs("pointconv_nosample")
    save("pointconv_nosample", out=m(T(x["xyz"]), T(x["feat"])).numpy())

    c = K.BLOCK_CASES["correlation3d"]
    m = ref_3d.Correlation3D(c["C"], c["C"], k=c["k"])
    load_params(m, c["seed"] + 1000)
    x = K.block_inputs("correlation3d")
    save("correlation3d", out=m(T(x["xyz1"]), T(x["feat1"]), T(x["xyz2"]), T(x["feat2"])).numpy())

    c = K.BLOCK_CASES["flow_estimator3d"]
    m = ref_3d.FlowEstimator3D(c["channels"], k=c["k"])
    load_params(m, c["seed"] + 1000)
    x = K.block_inputs("flow_estimator3d")
    knn = ref_ops.k_nearest_neighbor(T(x["xyz"]), T(x["xyz"]), k=c["k"])
    feat, flow = m(T(x["xyz"]), T(x["feat"]), knn)
    save("flow_estimator3d", feat=feat.numpy(), flow=flow.numpy())

    c = K.BLOCK_CASES["feature_pyramid3d"]
    m = ref_3d.FeaturePyramid3D(c["channels"], norm=c["norm"], k=c["k"])
    load_params(m, c["seed"] + 1000)
    x = K.block_inputs("feature_pyramid3d")
    xyzs1, xyzs2, idx1, idx2 = ref_3d.build_pc_pyramid(T(x["pc1"]), T(x["pc2"]), c["samples"])
    feats = m(xyzs1)
    save("feature_pyramid3d", **{"feat%d" % i: f.numpy() for i, f in enumerate(feats)},
         **{"index1_%d" % i: v.numpy() for i, v in enumerate(idx1)}, **{"index2_%d" % i: v.numpy() for i, v in enumerate(idx2)})


def gen_blocks_general():
    """The PointConv / Correlation3D modules outside the fused kernels' configuration (tests/cases.py GENERAL_CASES)."""
    out = {}
    for name, c in K.GENERAL_CASES.items():
        x = K.block_inputs(name)
        if c["kind"] == "corr":
            m = ref_3d.Correlation3D(c["C"], c["Cout"], k=c["k"])
            load_params(m, c["seed"] + 1000)
            with torch.no_grad():
                out[name] = m(T(x["xyz1"]), T(x["feat1"]), T(x["xyz2"]), T(x["feat2"])).numpy()
            continue
        cls = ref_pc.PointConvDownSampling if c["kind"] == "down" else ref_pc.PointConvNoSampling
        m = cls(c["C"], c["Cout"], norm=c["norm"], activation=c["activation"], k=c["k"])
        load_params(m, c["seed"] + 1000)
        m.train(c["train"])
        args = (T(x["xyz"]), T(x["feat"]), T(x["sampled"])) if c["kind"] == "down" else (T(x["xyz"]), T(x["feat"]))
        if c["train"]:  # training-mode BatchNorm + the gradients of sum(out^2) w.r.t. the features and the linear weight
            feat = args[1].clone().requires_grad_(True)
            y = m(args[0], feat, *args[2:])
            (y * y).sum().backward()
            out[name] = y.detach().numpy()
            out[name + "__grad_feat"] = feat.grad.numpy()
            out[name + "__grad_linear"] = m.linear.weight.grad.numpy()
            out[name + "__running_mean"] = m.norm_fn.running_mean.numpy().copy()
        else:
            with torch.no_grad():
                out[name] = m(*args).numpy()
    save("general_modules", **out)


def reference_model():
    """The reference RPEFlow on CPU: things.yaml model section, MI noise drawn on the CPU (the reference
    hard-codes torch.cuda.FloatTensor in mutual_info.py:32,84,155,211; its output never reaches the flows)."""
    with contextlib.redirect_stdout(io.StringIO()):
        from models import mutual_info as mi
        from models.RPEFlow import RPEFlow
    for cls in (mi.Mutual_info_reg_2D, mi.Mutual_info_reg_2D_Event, mi.Mutual_info_reg_3D, mi.Mutual_info_reg_3D_Event):
        cls.reparametrize = lambda self, mu, logvar: torch.randn_like(mu) * logvar.mul(0.5).exp() + mu
    from rpeflow_amd.model import things_config
    # record what the reference's host-side IDS transform (RPEFlow.py:68-69) hands to FPS / KNN: its log() differs by an
    # ulp between CPU models (AVX2 / AVX-512 paths), so the goldens carry the transformed clouds and the GPU tests feed them
    import models.RPEFlow as ref_mod
    original = ref_mod.perspect2parallel.__wrapped__ if hasattr(ref_mod.perspect2parallel, "__wrapped__") else ref_mod.perspect2parallel
    IDS_LOG.clear()

    def recording(xyz, persp, paral):
        out = original(xyz, persp, paral)
        IDS_LOG.append(out.detach().numpy().copy())
        return out
    recording.__wrapped__ = original
    ref_mod.perspect2parallel = recording
    return RPEFlow(things_config())


IDS_LOG = []


def ids_clouds():
    assert len(IDS_LOG) == 2, "one forward = two transformed clouds"
    return dict(pc1_ids=IDS_LOG[0], pc2_ids=IDS_LOG[1])


def model_params(module):
    return I.model_params([(k, tuple(v.shape)) for k, v in module.state_dict().items()])


@torch.no_grad()
def gen_model():
    import json
    m = reference_model()
    keys = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]
    with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f)
    print("state_dict_keys.json:", len(keys), "entries")
    m.load_state_dict({k: T(v) for k, v in model_params(m).items()}, strict=True)
    m.eval()
    sample = I.frame_pair(1000, H=128, W=192, N=8192)
    batch = {k: T(v)[None] for k, v in sample.items()}
    # per-level intermediates: what RPEFlow_core.decode returns (RPEFlow_core.py:432), every pyramid level's up-sampled flows,
    # before RPEFlow.forward maps the 3-D ones back through the IDS transform
    levels, decode = {}, m.pwc_fusion_core.decode

    def recording(*a, **k):
        flows_2d, flows_3d, mi = decode(*a, **k)
        levels.update({"level%d_flow_2d" % i: f.numpy().copy() for i, f in enumerate(flows_2d)})
        levels.update({"level%d_flow_3d" % i: f.numpy().copy() for i, f in enumerate(flows_3d)})
        return flows_2d, flows_3d, mi
    m.pwc_fusion_core.decode = recording
    out = m(batch, is_Train=False)
    m.pwc_fusion_core.decode = decode
    f2, f3 = out["flow_2d"].numpy(), out["flow_3d"].numpy()
    assert np.isfinite(f2).all() and np.isfinite(f3).all()
    print("flow_2d |max|", np.abs(f2).max(), "flow_3d |max|", np.abs(f3).max(), {k: v.shape for k, v in levels.items()})
    save("model_128x192", flow_2d=f2, flow_3d=f3, **ids_clouds(), **levels)


@torch.no_grad()
def gen_model_dsec():
    """DSEC-shaped sample (conf/test/dsec.yaml: 480x640 frames -> resize_to_64x 512x640, sensor 16x20).
    flow_2d is stored on a stride-8 grid (the full map is 2.4 MB) together with the EPE scalars."""
    m = reference_model()
    m.load_state_dict({k: T(v) for k, v in model_params(m).items()}, strict=True)
    m.eval()
    sample = I.frame_pair(2000, H=480, W=640, N=8192, dsec=True)
    batch = {k: T(v)[None] for k, v in sample.items()}
    out = m(batch, is_Train=False)
    f2, f3 = out["flow_2d"].numpy(), out["flow_3d"].numpy()
    assert np.isfinite(f2).all() and np.isfinite(f3).all()
    epe2 = float(np.sqrt(((f2 - sample["flow_2d"][None, :2]) ** 2).sum(1)).mean())
    epe3 = float(np.sqrt(((f3 - sample["flow_3d"][None, :3]) ** 2).sum(1)).mean())
    print("dsec flow_2d |max|", np.abs(f2).max(), "flow_3d |max|", np.abs(f3).max(), "EPE", epe2, epe3)
    save("model_dsec_480x640", flow_2d_s8=f2[:, :, ::8, ::8].copy(), flow_3d=f3, epe2d=np.float64(epe2), epe3d=np.float64(epe3), **ids_clouds())


@torch.no_grad()
def gen_model_full():
    """BASELINE config 3 shape: one 544x960 frame pair + 8192 points through the reference on the CPU.
    flow_2d is stored on a stride-8 grid (the full map is 4 MB) together with the EPE scalars."""
    m = reference_model()
    m.load_state_dict({k: T(v) for k, v in model_params(m).items()}, strict=True)
    m.eval()
    sample = I.frame_pair(3000, H=544, W=960, N=8192)
    batch = {k: T(v)[None] for k, v in sample.items()}
    out = m(batch, is_Train=False)
    f2, f3 = out["flow_2d"].numpy(), out["flow_3d"].numpy()
    assert np.isfinite(f2).all() and np.isfinite(f3).all()
    epe2 = float(np.sqrt(((f2 - sample["flow_2d"][None, :2]) ** 2).sum(1)).mean())
    epe3 = float(np.sqrt(((f3 - sample["flow_3d"][None, :3]) ** 2).sum(1)).mean())
    print("full flow_2d |max|", np.abs(f2).max(), "flow_3d |max|", np.abs(f3).max(), "EPE", epe2, epe3)
    save("model_544x960", flow_2d_s8=f2[:, :, ::8, ::8].copy(), flow_3d=f3, epe2d=np.float64(epe2), epe3d=np.float64(epe3), **ids_clouds())


@torch.no_grad()
def gen_fblocks():
    """Section 8(f) rows at block level: the reference's CrossTransformerBlock2D/3D, convex_upsample, resize_flow2d."""
    with contextlib.redirect_stdout(io.StringIO()):
        from models import restormer_arch as ref_ra
    for name, c in K.FBLOCK_CASES.items():
        x = K.fblock_inputs(name)
        if name.startswith("cross_block"):
            cls = ref_ra.CrossTransformerBlock2D if "2d" in name else ref_ra.CrossTransformerBlock3D
            m = cls(dim=c["C"], num_heads=c["heads"])
            load_params(m, c["seed"] + 1000)
            out = m(T(x["x"]), T(x["y"])).numpy()
            st = c.get("stride", 1)
            save(name, out=np.ascontiguousarray(out[:, :, ::st, ::st] if out.ndim == 4 else out[:, :, ::st]))
        elif name.startswith("convex_upsample"):
            save(name, out=ref_utils.convex_upsample(T(x["flow"]), T(x["mask"]), scale_factor=c["scale"]).numpy())
        else:
            save(name, out=ref_utils.resize_flow2d(T(x["flow"]).clone(), c["th"], c["tw"]).numpy())


@torch.no_grad()
def gen_model_bench_dsec():
    """bench.py --config dsec: batch 3 of 480x640 DSEC-shaped frame pairs (seeds 2000..2002) through the reference on the CPU."""
    c = K.BENCH_CASE_DSEC
    m = reference_model()
    m.load_state_dict({k: T(v) for k, v in model_params(m).items()}, strict=True)
    m.eval()
    samples = [I.frame_pair(c["first_seed"] + i, H=c["H"], W=c["W"], N=c["N"], dsec=True) for i in range(c["B"])]
    batch = {k: torch.stack([T(s[k]) for s in samples]) for k in samples[0]}
    out = m(batch, is_Train=False)
    f2, f3 = out["flow_2d"].numpy(), out["flow_3d"].numpy()
    assert np.isfinite(f2).all() and np.isfinite(f3).all()
    t2, t3 = batch["flow_2d"].numpy()[:, :2], batch["flow_3d"].numpy()[:, :3]
    epe2 = float(np.sqrt(((f2 - t2) ** 2).sum(1)).mean())
    epe3 = float(np.sqrt(((f3 - t3) ** 2).sum(1)).mean())
    print("bench dsec flow_2d |max|", np.abs(f2).max(), "flow_3d |max|", np.abs(f3).max(), "EPE", epe2, epe3)
    save("model_bench_dsec_b3_480x640", flow_2d_s8=f2[:, :, ::8, ::8].copy(), flow_3d=f3, epe2d=np.float64(epe2), epe3d=np.float64(epe3),
         **ids_clouds())


@torch.no_grad()
def gen_model_bench():
    """The benched configuration itself (bench.py / BASELINE config 3): a batch of 4 synthetic 544x960 frame pairs
    (seeds 1000..1003) + 8192 points through the reference on the CPU, seeded parameters.  Stored: flow_2d on a stride-8
    grid, flow_3d, the batch EPEs against the synthetic targets, the clouds as the reference's host IDS produced them."""
    c = K.BENCH_CASE
    m = reference_model()
    m.load_state_dict({k: T(v) for k, v in model_params(m).items()}, strict=True)
    m.eval()
    samples = [I.frame_pair(c["first_seed"] + i, H=c["H"], W=c["W"], N=c["N"]) for i in range(c["B"])]
    batch = {k: torch.stack([T(s[k]) for s in samples]) for k in samples[0]}
    out = m(batch, is_Train=False)
    f2, f3 = out["flow_2d"].numpy(), out["flow_3d"].numpy()
    assert np.isfinite(f2).all() and np.isfinite(f3).all()
    t2, t3 = batch["flow_2d"].numpy()[:, :2], batch["flow_3d"].numpy()[:, :3]
    epe2 = float(np.sqrt(((f2 - t2) ** 2).sum(1)).mean())
    epe3 = float(np.sqrt(((f3 - t3) ** 2).sum(1)).mean())
    print("bench flow_2d |max|", np.abs(f2).max(), "flow_3d |max|", np.abs(f3).max(), "EPE", epe2, epe3)
    save("model_bench_b4_544x960", flow_2d_s8=f2[:, :, ::8, ::8].copy(), flow_3d=f3, epe2d=np.float64(epe2), epe3d=np.float64(epe3),
         **ids_clouds())


@torch.no_grad()
def gen_model_stress():
    """Parity of the COMPOSITION off the easy regime (round-3 review): a second seeded parameter fill (STRESS_MODEL_SEED) and
    large-motion samples (tests/inputs.py frame_pair_stress: rigid motion + N(0, 0.5^2), ~5-10 % of the points projecting
    outside the frame, zero-mask and NaN pixels in the 2-D target) through the reference on the CPU: 128x192 with every
    decoder level's flows, 544x960 (BASELINE config 3's frame) and a DSEC-shaped 480x640 with the masked EPEs."""
    m = reference_model()
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    m.load_state_dict({k: T(v) for k, v in I.model_params(shapes, seed=I.STRESS_MODEL_SEED).items()}, strict=True)
    m.eval()
    for name, seed, H, W, dsec in (("model_128x192_stress", 5000, 128, 192, False), ("model_544x960_stress", 5001, 544, 960, False),
                                   ("model_dsec_480x640_stress", 5002, 480, 640, True)):
        IDS_LOG.clear()
        sample = I.frame_pair_stress(seed, H=H, W=W, N=8192, dsec=dsec)
        batch = {k: T(v)[None] for k, v in sample.items()}
        levels, decode = {}, m.pwc_fusion_core.decode

        def recording(*a, **k):
            flows_2d, flows_3d, mi = decode(*a, **k)
            if H == 128:
                levels.update({"level%d_flow_2d" % i: f.numpy().copy() for i, f in enumerate(flows_2d)})
                levels.update({"level%d_flow_3d" % i: f.numpy().copy() for i, f in enumerate(flows_3d)})
            return flows_2d, flows_3d, mi
        m.pwc_fusion_core.decode = recording
        out = m(batch, is_Train=False)
        m.pwc_fusion_core.decode = decode
        f2, f3 = out["flow_2d"].numpy(), out["flow_3d"].numpy()
        assert np.isfinite(f2).all() and np.isfinite(f3).all(), name
        epe2, epe3 = I.masked_epes(f2[0], f3[0], sample)
        print(name, "flow_2d |max|", np.abs(f2).max(), "flow_3d |max|", np.abs(f3).max(), "EPE", epe2, epe3)
        save(name, **({"flow_2d": f2} if H == 128 else {"flow_2d_s8": f2[:, :, ::8, ::8].copy()}), flow_3d=f3,
             epe2d=np.float64(epe2), epe3d=np.float64(epe3), **ids_clouds(), **levels)


IDS_SWEEP = dict(first_seed=6000, pairs=32, H=544, W=960, N=8192, samples=4096)


@torch.no_grad()
def gen_ids_sweep():
    """What the product's DEFAULT path replaces -- the host-side IDS transform (RPEFlow.py:56-69 -> utils.py:320-346) followed by
    build_pc_pyramid's furthest-point sampling (pwc3d_core.py:11-13) -- recorded from the reference for 64 clouds (32 stress
    frame pairs, seeds 6000...): per cloud the full sampling order (uint16) and, instead of the 6 MB of transformed clouds,
    the positions and values where the reference's z' (torch.log on this container's CPU) is NOT the correctly rounded one
    of oracle.perspect2parallel; x' and y' are asserted bit-identical here, so the reference cloud is oracle + patches."""
    from oracle import oracle as O
    c = IDS_SWEEP
    H, W, N = c["H"], c["W"], c["N"]
    Hp, Wp = (H + 63) // 64 * 64 // 32, (W + 63) // 64 * 64 // 32
    orders, patch_cloud, patch_pos, patch_val = [], [], [], []
    for i in range(c["pairs"]):
        sample = I.frame_pair_stress(c["first_seed"] + i, H=H, W=W, N=N)
        pcs, intr = T(sample["pcs"])[None], T(sample["intrinsics"])[None]
        persp = {"projection_mode": "perspective", "sensor_h": H, "sensor_w": W, "f": intr[:, 0], "cx": intr[:, 1], "cy": intr[:, 2]}
        paral = {"projection_mode": "parallel", "sensor_h": Hp, "sensor_w": Wp, "cx": (Wp - 1) / 2, "cy": (Hp - 1) / 2}
        clouds = [ref_utils.perspect2parallel(pcs[:, sl], persp, paral) for sl in (slice(0, 3), slice(3, 6))]
        both = torch.cat(clouds, dim=0)
        order = ref_ops.furthest_point_sampling(both.transpose(1, 2), c["samples"]).numpy()
        for j in range(2):
            mine = O.perspect2parallel(sample["pcs"][None, 3 * j:3 * j + 3], sample["intrinsics"][None], H, W, Hp, Wp)[0]
            ref = clouds[j][0].numpy()
            assert np.array_equal(mine[:2].view(np.uint32), ref[:2].view(np.uint32))
            off = np.nonzero(mine[2].view(np.uint32) != ref[2].view(np.uint32))[0]
            patch_cloud += [2 * i + j] * len(off)
            patch_pos += off.tolist()
            patch_val += ref[2][off].tolist()
            orders.append(order[j].astype(np.uint16))
        print("pair", i, "z' values off the correctly rounded log so far:", len(patch_pos), flush=True)
    save("ids_fps_sweep", order=np.stack(orders), patch_cloud=np.array(patch_cloud, np.int32), patch_pos=np.array(patch_pos, np.int32),
         patch_val=np.array(patch_val, np.float32))


def gen_eval():
    """The reference's evaluation loops themselves (eval_withocc.py:45-135, eval_noocc.py:45-116), unmodified, over the
    synthetic frame pairs and the stand-in predictions of tests/test_evaluate.py; stored: the accumulated metric sums.
    The scripts import their whole project at module level -- datasets, visualisation, configuration -- through modules
    that are not installed here (cv2, imageio, h5py, omegaconf) and that ``Evaluator.run`` never touches: empty placeholder
    modules satisfy those import statements and ``factory`` (dataset / model construction, unused by ``run``) is a
    placeholder as a whole.  ``Evaluator.__init__`` (dataset, checkpoint) is bypassed: the object gets a list of batches
    and a stand-in model.  torch.cuda.synchronize() is a no-op for the CPU run.  The metric dictionaries are locals of
    ``run``; a profile hook copies them when the function returns."""
    import types
    from rpeflow_amd.evaluate import collate
    from rpeflow_amd.synthetic import SyntheticPairs
    from tests.test_evaluate import fake_model
    for name in ("cv2", "imageio", "h5py"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    if "omegaconf" not in sys.modules:
        omegaconf = types.ModuleType("omegaconf")
        omegaconf.DictConfig = dict
        sys.modules["omegaconf"] = omegaconf
    factory = types.ModuleType("factory")
    factory.model_factory = factory.dataset_factory = None
    sys.modules["factory"] = factory

    class StandIn:
        def eval(self):
            return self

        def forward(self, inputs, is_Train=False):
            return fake_model(inputs)

    def run(module_name, dsec):
        module = __import__(module_name)
        data = SyntheticPairs(5, H=24, W=40, N=512, dsec=dsec)
        ev = object.__new__(module.Evaluator)
        ev.device, ev.cfgs, ev.model = torch.device("cpu"), None, StandIn()
        ev.test_loader = [collate([data[i] for i in idx]) for idx in ((0, 1), (2, 3), (4,))]
        got = {}

        def hook(frame, event, arg):
            if event == "return" and frame.f_code.co_name == "run" and frame.f_code.co_filename.endswith(module_name + ".py"):
                got.update({k: dict(v) for k, v in frame.f_locals.items() if k.startswith("metrics_")})
        sync, torch.cuda.synchronize = torch.cuda.synchronize, (lambda *a, **k: None)
        sys.setprofile(hook)
        try:
            with contextlib.redirect_stderr(io.StringIO()):  # tqdm
                ev.run()
        finally:
            sys.setprofile(None)
            torch.cuda.synchronize = sync
        return got

    m = run("eval_withocc", dsec=False)
    withocc = [m["metrics_2d"][k] for k in ("counts", "EPE2d", "1px", "Fl")] + [m["metrics_3d"][k] for k in ("counts", "EPE3d", "5cm", "10cm")] + [
        m["metrics_3d_noc"][k] for k in ("counts", "EPE3d", "5cm", "10cm")]
    m = run("eval_noocc", dsec=True)
    noocc = [m["metrics_2d"][k] for k in ("counts", "EPE2d", "1px", "Fl")] + [m["metrics_3d"][k] for k in ("counts", "EPE3d", "5cm", "10cm")]
    print("withocc", withocc, "\nnoocc", noocc)
    save("eval_accumulators", withocc=np.asarray(withocc, np.float64), noocc=np.asarray(noocc, np.float64))


# ------------------------------------------------------------------ the reference model's own calls
TRACE_CASES = {
    # the model goldens' regime: standard parameter fill, pc2 = pc1 + small noise, every point inside the frame
    "call_trace": dict(B=2, H=128, W=192, N=8192, first_seed=1000, stress=False, model_seed=None),
    # off the easy regime (tests/inputs.py frame_pair_stress): the second parameter fill, a rigid motion + N(0, 0.5^2), ~5 % of the
    # points projecting outside the frame -- the cross-cloud searches are no longer near-self searches, grid_sample_wrapper's
    # zero padding and the nearest-pixel search off the raster take part
    "call_trace_stress": dict(B=1, H=128, W=192, N=8192, first_seed=5000, stress=True, model_seed="stress"),
}
# functions wrapped where the reference's modules bound them at import time (``from .csrc import ...``): (module, name) pairs
TRACE_FUNCTIONS = {
    "k_nearest_neighbor": ["RPEFlow_core", "pwc3d_core", "pointconv", "utils"],
    "furthest_point_sampling": ["pwc3d_core"],
    "correlation2d": ["RPEFlow_core"],
    "batch_indexing_channel_first": ["utils", "pwc3d_core", "pointconv"],
    "batch_indexing_channel_last": ["utils", "pointconv"],
    "knn_interpolation": ["utils", "RPEFlow_core"],
    "backwarp_3d": ["RPEFlow_core"],
    "backwarp_2d": ["RPEFlow_core"],
    "grid_sample_wrapper": ["utils", "RPEFlow_core"],
    "project_feat_with_nn_corr": ["RPEFlow_core"],
    "build_pc_pyramid": ["RPEFlow"],
}
TRACE_CLASSES = {"pointconv": ["PointConvDownSampling", "PointConvNoSampling"], "pwc3d_core": ["Correlation3D", "FlowEstimator3D", "FeaturePyramid3D"]}
TRACE_GEOMETRY = {"input_xyz", "query_xyz", "xyz", "xyz1", "xyz2", "sampled_xyz", "xy", "flow12", "pc1", "pc2", "xyzs"}


class _Recorder:
    """Records calls (see tests/trace_io.py for the format); one instance per generated trace."""

    def __init__(self):
        import hashlib
        self.sha = hashlib.sha1
        self.calls, self.arrays, self.by_hash = [], {}, {}
        self.enabled, self.stack, self.module_names = False, [], {}

    # -- storage
    def store(self, a):
        a = np.ascontiguousarray(a)
        if a.dtype == np.int64:  # indices: narrowed for storage, widened again on load
            a = a.astype(np.uint16 if (a.size == 0 or (a.min() >= 0 and a.max() < 65536)) else np.int32)
        h = self.sha(str((a.dtype, a.shape)).encode() + a.tobytes()).hexdigest()
        if h not in self.by_hash:
            self.by_hash[h] = "a%04d" % len(self.arrays)
            self.arrays[self.by_hash[h]] = a
        return self.by_hash[h]

    def describe(self, t, name, seed):
        from tests import trace_io as TIO
        rec = dict(shape=list(t.shape), dtype=str(t.dtype).replace("torch.", ""), strides=list(t.stride()), offset=int(t.storage_offset()))
        v = t.detach().contiguous().numpy()
        geometry = name in TRACE_GEOMETRY or not t.dtype.is_floating_point or (t.dim() == 3 and min(t.shape[1], t.shape[2]) <= 3 and t.numel() <= 65536)
        if geometry or t.numel() * t.element_size() <= TIO.FULL_BYTES:
            rec["key"] = self.store(v)
        else:
            scale, shift = float(np.float32(v.std())), float(np.float32(v.mean()))
            rec["synthetic"] = dict(seed=seed, scale=scale, shift=shift)
        return rec

    def describe_output(self, out, seed):
        from tests import trace_io as TIO
        if isinstance(out, (list, tuple)):
            return dict(kind="list", items=[self.describe_output(o, seed * 16 + i) for i, o in enumerate(out)])
        rec = dict(kind="tensor", shape=list(out.shape), dtype=str(out.dtype).replace("torch.", ""))
        v = out.detach().contiguous().numpy()
        small_geometry = out.dim() == 3 and out.shape[1] <= 3 and out.numel() <= 65536
        if not out.dtype.is_floating_point or out.numel() <= TIO.SAMPLE_ABOVE or small_geometry:
            rec["key"] = self.store(v)
        else:
            rec["sampled"] = dict(seed=seed, n=TIO.N_SAMPLES)
            rec["key"] = self.store(v.reshape(-1)[TIO.sample_positions(seed, v.size)])
        return rec

    # -- one call
    def begin(self, fn_name, site, params, module=None):
        """``params``: [(name, value, "pos" | "kw")] in call order."""
        index = len(self.calls)
        call = dict(index=index, fn=fn_name, site=site, parent=self.stack[-1] if self.stack else None, args=[])
        if module is not None:
            call["module"] = module
        seen = []
        for j, (name, value, passed) in enumerate(params):
            a = dict(name=name, passed=passed)
            if torch.is_tensor(value):
                a["kind"] = "tensor"
                same = [n for n, v in seen if v is value]
                if same:
                    a["same_as"] = same[0]
                else:
                    a["t"] = self.describe(value, name, seed=1_000_000 + index * 16 + j)
                seen.append((name, value))
            elif isinstance(value, (list, tuple)) and value and all(torch.is_tensor(v) for v in value):
                a["kind"] = "tensor_list"
                a["t"] = [self.describe(v, name, seed=2_000_000 + index * 64 + j * 8 + i) for i, v in enumerate(value)]
            elif value is None:
                a["kind"] = "none"
            else:
                a["kind"] = "value"
                a["value"] = list(value) if isinstance(value, (list, tuple)) else value
            call["args"].append(a)
        self.calls.append(call)
        self.stack.append(index)
        return call

    def finish(self, call, rerun, out):
        """``rerun(args, kwargs)`` calls the reference function again; used when an argument was replaced by seeded values."""
        from tests import trace_io as TIO
        self.stack.pop()
        synthetic = [a["name"] for a in call["args"] if a["kind"] == "tensor" and "t" in a and "synthetic" in a["t"]]
        if synthetic:
            call["synthetic_args"] = synthetic
            trace = TIO.Trace.__new__(TIO.Trace)
            trace.arrays = self.arrays
            args, kwargs = TIO.Trace.arguments(trace, call, "cpu")
            was, self.enabled = self.enabled, False
            try:
                out = rerun(args, kwargs)
            finally:
                self.enabled = was
        call["out"] = self.describe_output(out, seed=3_000_000 + call["index"])


def _site(skip_prefixes=("torch",)):
    """file:line of the nearest frame that is reference code (models/...)."""
    f = sys._getframe(2)
    while f is not None:
        name = f.f_code.co_filename
        if name.startswith(REF) and not name.endswith("make_golden.py"):
            return os.path.relpath(name, REF) + ":%d" % f.f_lineno
        f = f.f_back
    return "?"


def _module_record(rec, mod):
    cls = type(mod).__name__
    norm = lambda m: {"BatchNorm1d": "batch_norm", "InstanceNorm1d": "instance_norm", "Identity": None}[type(m.norm_fn).__name__]
    act = lambda m: {"ReLU": "relu", "LeakyReLU": "leaky_relu", "Identity": None}[type(m.activation_fn).__name__]
    if cls.startswith("PointConv"):
        ctor = dict(in_channels=mod.linear.in_features // 16 - 3, out_channels=mod.linear.out_features, norm=norm(mod), activation=act(mod), k=mod.k)
    elif cls == "Correlation3D":
        first = mod.cost_mlp.convs[0].conv_fn
        ctor = dict(in_channels=(first.in_channels - 3) // 2, out_channels=first.out_channels, k=mod.k)
    elif cls == "FlowEstimator3D":
        ctor = dict(n_channels=[mod.point_conv1.linear.in_features // 16 - 3, mod.point_conv1.linear.out_features, mod.point_conv2.linear.out_features,
                                mod.mlp.convs[-1].conv_fn.out_channels], norm=norm(mod.point_conv1), conv_last=mod.conv_last is not None, k=mod.point_conv1.k)
    else:  # FeaturePyramid3D
        chans = [mod.level0_mlp.convs[-1].conv_fn.out_channels] + [m.convs[-1].conv_fn.out_channels for m in mod.pyramid_mlps]
        ctor = dict(n_channels=chans, norm=norm(mod.pyramid_convs[0]), k=mod.pyramid_convs[0].k)
    return dict(name=rec.module_names[id(mod)], cls=cls, ctor=ctor)


def gen_call_trace():
    for name in TRACE_CASES:
        _gen_call_trace(name)


@torch.no_grad()
def _gen_call_trace(trace_name):
    """Every call the reference MODEL makes into the hot path during one forward (128 x 192 frames, 8192 points, seeded
    parameters; TRACE_CASES): the four names of models/csrc, the section-8(a) glue functions of models/utils.py, build_pc_pyramid and the
    PointConv / Correlation3D / FlowEstimator3D / FeaturePyramid3D forwards -- how each was called (positional / keyword),
    shapes, dtypes, strides, storage offsets, values and outputs.  Format and storage policy: tests/trace_io.py."""
    import importlib
    import inspect
    import json
    from tests import trace_io as TIO

    c = TRACE_CASES[trace_name]
    m = reference_model()
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    seed_kw = {"seed": I.STRESS_MODEL_SEED} if c["model_seed"] == "stress" else {}
    m.load_state_dict({k: T(v) for k, v in I.model_params(shapes, **seed_kw).items()}, strict=True)
    m.eval()
    rec = _Recorder()
    rec.module_names = {id(mod): name for name, mod in m.named_modules()}
    undo = []

    def wrap_function(fn_name, original):
        sig = inspect.signature(original)

        def wrapper(*args, **kwargs):
            if not rec.enabled:
                return original(*args, **kwargs)
            names = list(sig.parameters)
            params = [(names[i], v, "pos") for i, v in enumerate(args)] + [(k, v, "kw") for k, v in kwargs.items()]
            call = rec.begin(fn_name, _site(), params)
            out = original(*args, **kwargs)
            rec.finish(call, lambda a, k: original(*a, **k), out)
            return out
        return wrapper

    def wrap_forward(cls):
        original = cls.forward
        sig = inspect.signature(original)

        def forward(self, *args, **kwargs):
            if not rec.enabled:
                return original(self, *args, **kwargs)
            names = list(sig.parameters)[1:]
            params = [(names[i], v, "pos") for i, v in enumerate(args)] + [(k, v, "kw") for k, v in kwargs.items()]
            call = rec.begin(cls.__name__ + ".forward", _site(), params, module=_module_record(rec, self))
            out = original(self, *args, **kwargs)
            rec.finish(call, lambda a, k: original(self, *a, **k), out)
            return out
        cls.forward = forward
        undo.append(lambda: setattr(cls, "forward", original))

    for fn_name, modules in TRACE_FUNCTIONS.items():
        wrapped = None
        for mod_name in modules:
            mod = importlib.import_module("models." + mod_name)
            original = getattr(mod, fn_name)
            wrapped = wrapped or wrap_function(fn_name, original)
            setattr(mod, fn_name, wrapped)
            undo.append(lambda mod=mod, fn_name=fn_name, original=original: setattr(mod, fn_name, original))
    for mod_name, classes in TRACE_CLASSES.items():
        for cls_name in classes:
            wrap_forward(getattr(importlib.import_module("models." + mod_name), cls_name))

    make = I.frame_pair_stress if c["stress"] else I.frame_pair
    samples = [make(c["first_seed"] + i, H=c["H"], W=c["W"], N=c["N"]) for i in range(c["B"])]
    batch = {k: torch.stack([T(s[k]) for s in samples]) for k in samples[0]}
    rec.enabled = True
    try:
        out = m(batch, is_Train=False)
    finally:
        rec.enabled = False
        for u in reversed(undo):
            u()
    assert torch.isfinite(out["flow_2d"]).all() and torch.isfinite(out["flow_3d"]).all()

    counts = {}
    for call in rec.calls:
        counts[call["fn"]] = counts.get(call["fn"], 0) + 1
    print("calls:", len(rec.calls), counts)
    meta = dict(case=c, parameters="tests.inputs.model_params over tests/golden/state_dict_keys.json" + (" with seed STRESS_MODEL_SEED" if seed_kw else ""),
                full_bytes=TIO.FULL_BYTES,
                sample_above=TIO.SAMPLE_ABOVE, n_samples=TIO.N_SAMPLES, counts=counts, calls=rec.calls)
    with open(os.path.join(OUT, trace_name + ".json"), "w") as f:
        json.dump(meta, f, separators=(",", ":"))
    save(trace_name, **rec.arrays)
    print("%s.json: %.0f KiB, %d arrays" % (trace_name, os.path.getsize(os.path.join(OUT, trace_name + ".json")) / 1024, len(rec.arrays)))


if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1:] or ["knn", "fps", "sqdist", "corr", "glue", "blocks", "model", "model_dsec", "model_full", "events", "eval", "fblocks", "model_bench", "model_bench_dsec", "model_stress", "blocks_general", "ids_sweep", "call_trace"]
    for w in which:
        globals()["gen_" + w]()
