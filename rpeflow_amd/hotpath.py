"""The hot-path operator sequence of ONE RPEFlow forward, at the shapes the model
produces for a batch of FlyingThings3D-sized frame pairs.

This is the bench workload: every call the reference's forward makes into the
hot path -- FPS, the 43 KNNs, correlation2d, backwarp_2d/3d, knn_interpolation,
project_feat_with_nn_corr, grid_sample_wrapper, the PointConv layers of
FeaturePyramid3D / FlowEstimator3D and Correlation3D -- in the reference's order
and with its tensor shapes (RPEFlow.py:36-99, RPEFlow_core.py:302-432).  The dense
2-D convolutions, Restormer attention and MI heads between those calls are NOT part
of the hot path and are replaced by resident synthetic
feature maps of the right shape, so a step times the hot path and nothing else.
"""
import contextlib

import torch

from types import SimpleNamespace

OP_NAMES = ["correlation2d", "k_nearest_neighbor", "build_pc_pyramid", "FeaturePyramid3D", "Correlation3D",
            "FlowEstimator3D", "backwarp_2d", "backwarp_3d", "grid_sample_wrapper", "knn_interpolation",
            "project_feat_with_nn_corr"]


def native_ops():
    """The HIP-backed implementations of this package (the only ones it ships)."""
    from . import csrc, pwc3d_core, utils
    table = {}
    for name in OP_NAMES:
        for mod in (csrc, pwc3d_core, utils):
            if hasattr(mod, name):
                table[name] = getattr(mod, name)
                break
    return SimpleNamespace(**table)

PYRAMID_2D = [16, 32, 64, 96, 128, 192]   # RPEFlow_core.py:174-177
PYRAMID_3D = [16, 32, 64, 96, 128, 192]   # RPEFlow_core.py:215-219
N_SAMPLES = [4096, 2048, 1024, 512, 256]  # RPEFlow.py:74


class Timer:
    """Per-category GPU time, measured with events on the stream the kernels run on."""

    def __init__(self, enabled):
        self.enabled = enabled
        self.spans = {}

    @contextlib.contextmanager
    def span(self, name):
        if not self.enabled:
            yield
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        yield
        b.record()
        self.spans.setdefault(name, []).append((a, b))

    def totals_ms(self):
        torch.cuda.synchronize()
        return {k: (sum(a.elapsed_time(b) for a, b in v), len(v)) for k, v in self.spans.items()}


class SegmentGraphs(Timer):
    """The step as a sequence of HIP graphs, one per span.

    capture(): runs the workload once with every span body recorded into its own graph (all graphs
    share one memory pool and are replayed in capture order, which is what makes that sharing
    legal).  replay(): launches the graphs back to back, bracketing each with events on the launch
    stream -- so a step costs ~70 graph launches instead of ~600 kernel launches plus the Python
    between them, and the per-category times are still measured live."""

    def __init__(self):
        super().__init__(True)
        self.segments = []
        self.pool = torch.cuda.graph_pool_handle()
        self.capturing = False

    @contextlib.contextmanager
    def span(self, name):
        if not self.capturing:
            raise RuntimeError("SegmentGraphs.span outside capture()")
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self.pool, capture_error_mode="thread_local"):
            yield
        self.segments.append((name, g))

    def capture(self, workload):
        self.capturing = True
        try:
            self.outputs = workload(self)
        finally:
            self.capturing = False
        torch.cuda.synchronize()
        return self.outputs

    def replay(self, timed=True):
        for name, g in self.segments:
            if timed:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                g.replay()
                b.record()
                self.spans.setdefault(name, []).append((a, b))
            else:
                g.replay()
        return self.outputs


class HotPathWorkload(torch.nn.Module):
    def __init__(self, batch=4, height=544, width=960, n_points=8192, device="cuda:0", seed=0, ops=None):
        """``ops``: a namespace with the OP_NAMES callables/classes; default = this package's
        HIP-backed ones.  (bench.py passes a CPU port here for its cpu_baseline leg.)"""
        super().__init__()
        self.ops = ops = ops or native_ops()
        self.B, self.N = batch, n_points
        g = torch.Generator(device="cpu").manual_seed(seed)
        H64, W64 = (height + 63) // 64 * 64, (width + 63) // 64 * 64  # resize_to_64x, utils.py:227-241
        self.sensor = (H64 // 32, W64 // 32)  # ids.sensor_size_divisor = 32 (conf/test/things.yaml)
        self.sizes = [(H64 >> (l + 1), W64 >> (l + 1)) for l in range(6)]

        def rnd(*shape, scale=1.0):
            return (torch.randn(*shape, generator=g) * scale).to(device)

        # clouds as the IDS transform leaves them (utils.py:320-346): x,y in sensor pixels, z = f*log(depth)-ish
        sh, sw = self.sensor
        xy = torch.rand(batch, 2, n_points, generator=g) * torch.tensor([sw - 1.0, sh - 1.0])[None, :, None]
        xy = xy - torch.tensor([(sw - 1) / 2, (sh - 1) / 2])[None, :, None]
        z = 22.0 + 91.0 * torch.rand(batch, 1, n_points, generator=g)
        pc1 = torch.cat([xy, z], 1).to(device)
        self.pc_both = torch.cat([pc1, pc1 + rnd(batch, 3, n_points, scale=0.05)], 0).contiguous()  # [2B,3,N] as the IDS transform returns the clouds
        self.pc1, self.pc2 = self.pc_both[:batch], self.pc_both[batch:]

        self.feats_2d_both = [rnd(2 * batch, c, h, w) for c, (h, w) in zip(PYRAMID_2D, self.sizes)]  # frame 1 then frame 2
        self.feats1_2d = [f[:batch] for f in self.feats_2d_both]
        self.feats2_2d = [f[batch:] for f in self.feats_2d_both]
        self.efeats_2d = [rnd(batch, c, h, w) for c, (h, w) in zip([32, 32, 64, 96, 128, 192], self.sizes)]
        self.flow_feat_2d = [rnd(batch, 64, h, w) for (h, w) in self.sizes]  # flow_estimator_2d.flow_feat_dim = 32+32
        self.flow_2d = [rnd(batch, 2, h, w, scale=2.0) for (h, w) in self.sizes]

        # per-level constants of the projection (RPEFlow_core.py:316-330), resident like the inputs
        self.scale, self.centre, self.grid = [], [], []
        for (h, w) in self.sizes:
            self.scale.append(torch.tensor([(w - 1) / (sw - 1), (h - 1) / (sh - 1)])[None, :, None].to(device))
            self.centre.append(torch.tensor([(sw - 1) / 2, (sh - 1) / 2])[None, :, None].to(device))
            gx, gy = torch.arange(w, dtype=torch.float32), torch.arange(h, dtype=torch.float32)
            grid = torch.stack([gx[None, :].expand(h, w), gy[:, None].expand(h, w)]).reshape(1, 2, -1)
            self.grid.append(grid.to(device).expand(batch, 2, h * w))
        self.grid_both = [g[:1].expand(2 * batch, 2, g.shape[2]) for g in self.grid]
        self.camera = {"projection_mode": "parallel", "cx": (sw - 1) / 2, "cy": (sh - 1) / 2, "sensor_h": sh, "sensor_w": sw}
        self.scale_xy = [((w - 1) / (sw - 1), (h - 1) / (sh - 1)) for (h, w) in self.sizes]
        self.zero_flow_3d = torch.zeros(batch, 3, N_SAMPLES[-1], device=device)
        self.zero_flow_feat_3d = torch.zeros(batch, 64, N_SAMPLES[-1], device=device)

        torch.manual_seed(seed)
        self.feature_pyramid_3d = ops.FeaturePyramid3D(PYRAMID_3D, norm="batch_norm", k=16)
        self.correlations_3d = torch.nn.ModuleList(
            [torch.nn.Identity()] + [ops.Correlation3D(c, c, k=16) for c in PYRAMID_3D[1:]])
        self.flow_estimator_3d = ops.FlowEstimator3D([64 + 64 + 3 + 64, 128, 128, 64], None, conv_last=False, k=16)
        self.flow_head_3d = torch.nn.Conv1d(64, 3, kernel_size=1)
        from .utils import Conv1dNormRelu  # feature_aligners_3d / correlation_aligners_3d are Conv1dNormRelu(c, 64) (RPEFlow_core.py:220-242)
        self.aligners = torch.nn.ModuleList([Conv1dNormRelu(c, 64) for c in PYRAMID_3D])
        self.to(device).eval()

    @torch.no_grad()
    def forward(self, timer=None):
        t = timer or Timer(False)
        B, o = self.B, self.ops
        correlation2d, k_nearest_neighbor, build_pc_pyramid = o.correlation2d, o.k_nearest_neighbor, o.build_pc_pyramid
        backwarp_2d, backwarp_3d, grid_sample_wrapper = o.backwarp_2d, o.backwarp_3d, o.grid_sample_wrapper
        knn_interpolation, project_feat_with_nn_corr = o.knn_interpolation, o.project_feat_with_nn_corr
        import inspect
        shares = "sampled_2d" in inspect.signature(project_feat_with_nn_corr).parameters
        if shares:
            from .utils import conv_module, grid_sample_sources, project_points
        sh, sw = self.sensor
        # Frames 1 and 2 go through the shared-weight 3-D pyramid and the pyramid fusers' hot-path calls as ONE batch of 2B
        # clouds / maps, as rpeflow_amd.model does (the reference calls them once per frame, RPEFlow.py:78-79,
        # RPEFlow_core.py:329-337; per-sample operators make the two forms equal sample by sample).  With the native operators
        # (``shares``) the calls are the ones rpeflow_amd.model makes: the glue the reference puts between its operators
        # (concatenations, scalings, the flow heads' adds) rides inside the launches; a CPU port passed as ``ops`` keeps the
        # reference's calls.
        with t.span("fps+pyramid"):
            if shares:
                xyzs1, xyzs2, _, _, xyzs_both = build_pc_pyramid(self.pc1, self.pc2, N_SAMPLES, return_both=True)
            else:
                xyzs1, xyzs2, _, _ = build_pc_pyramid(self.pc1, self.pc2, N_SAMPLES)
                xyzs_both = [torch.cat([a, b], 0) for a, b in zip(xyzs1, xyzs2)]
        with t.span("feature_pyramid_3d"):
            feats_both = self.feature_pyramid_3d(xyzs_both)
            feats1_3d, feats2_3d = [f[:B] for f in feats_both], [f[B:] for f in feats_both]

        flows_3d, flow_feats_3d = [], []
        interp_knn = {}  # level l -> the 3 nearest points of level l + 1 for every point of level l
        for level in range(5, 0, -1):
            xyz1, xyz2 = xyzs1[level], xyzs2[level]
            f1_2d, f2_2d, ef_2d = self.feats1_2d[level], self.feats2_2d[level], self.efeats_2d[level]
            f1_3d, f2_3d = feats1_3d[level], feats2_3d[level]
            h, w = self.sizes[level]
            n = xyz1.shape[-1]
            scale, centre, grid = self.scale[level], self.centre[level], self.grid[level]
            f_2d_both, f_3d_both = self.feats_2d_both[level], feats_both[level]
            # project_pc2image 'parallel' + rescale (RPEFlow_core.py:316-324): one launch with the native operators, tensor ops
            # ("torch_glue") with a port of the reference
            with t.span("project_pc2image" if shares else "torch_glue"):
                if shares:
                    xy_both = project_points(xyz1, xyz2, self.camera, self.scale_xy[level][0], self.scale_xy[level][1])
                else:
                    xy_both = (torch.cat([xyz1[:, :2], xyz2[:, :2]], 0) + centre) * scale
                xy1 = xy_both[:B]

            with t.span("knn2d_k1"):
                nn_proj_both = k_nearest_neighbor(xy_both, self.grid_both[level], k=1)
                nn_proj1 = nn_proj_both[:B]
            with t.span("knn3d_k16"):
                knn_1in1 = k_nearest_neighbor(xyz1, xyz1, k=16)

            # The 2-D and the 3-D fuser of a pair sample the same map at the same points (RPEFlow_core.py:334-337): the 3-D
            # fuser's grid_sample_wrapper goes first and the 2-D fuser takes its result instead of repeating the taps
            # (``shares``: the native operators; a CPU port of the reference passed as ``ops`` keeps the reference's calls).
            with t.span("grid_sample"):  # pyramid fusers 3D (:336-337)
                sampled = grid_sample_wrapper(f_2d_both, xy_both)
            with t.span("project_feat"):  # pyramid fusers 2D (:334-335)
                project_feat_with_nn_corr(xy_both, f_2d_both, f_3d_both, nn_proj_both[..., 0], **({"sampled_2d": sampled} if shares else {}))

            if level == 5:
                if shares:  # constants, as rpeflow_amd.model keeps them
                    last_flow_3d, last_flow_feat_3d = self.zero_flow_3d, self.zero_flow_feat_3d
                else:
                    with t.span("torch_glue"):
                        last_flow_3d = torch.zeros(B, 3, n, device=xyz1.device)
                        last_flow_feat_3d = torch.zeros(B, 64, n, device=xyz1.device)
                xyz2_warp, f2_2d_warp = xyz2, f2_2d
            else:
                with t.span("backwarp_2d"):
                    f2_2d_warp = backwarp_2d(f2_2d, self.flow_2d[level], padding_mode="border")
                with t.span("knn_interpolation"):
                    if shares:  # the same two clouds are interpolated again in the final up-sampling below: one search
                        up, interp_knn[level] = knn_interpolation(xyzs1[level + 1], (flows_3d[-1], flow_feats_3d[-1]), xyz1, return_indices=True)
                    else:
                        up = knn_interpolation(xyzs1[level + 1], torch.cat([flows_3d[-1], flow_feats_3d[-1]], 1), xyz1)
                    last_flow_3d, last_flow_feat_3d = up[:, :3], up[:, 3:]
                with t.span("backwarp_3d"):
                    xyz2_warp = backwarp_3d(xyz1, xyz2, last_flow_3d)

            with t.span("correlation3d"):
                corr_3d = self.correlations_3d[level](xyz1, f1_3d, xyz2_warp, f2_3d, knn_1in1)
            with t.span("correlation2d"):
                if shares:  # the native operators: the activation of RPEFlow_core.py:362 in the kernel's epilogue, as rpeflow_amd.model calls it
                    from .csrc.wrapper import _correlation2d_algo
                    corr_2d = _correlation2d_algo(f1_2d, f2_2d_warp, 4, 0, leaky_slope=0.1)
                else:
                    corr_2d = torch.nn.functional.leaky_relu(correlation2d(f1_2d, f2_2d_warp, 4), 0.1)

            if shares:
                with t.span("grid_sample"):  # corr fuser 3D (:376; RPEFlow_core.py:103-111 in one launch)
                    to_sensor = ((sw - 1, w - 1), (sh - 1, h - 1))  # the 2-D flow in sensor units: "* (sensor_w - 1) / (image_w - 1)", two roundings (:367-370)
                    sampled = grid_sample_sources([(corr_2d, None, None), (self.flow_2d[level], to_sensor, last_flow_3d[:, :2]), (ef_2d, None, None)], xy1)
                with t.span("project_feat"):  # corr fuser 2D (:371-373, :82-83)
                    project_feat_with_nn_corr(xy1, corr_2d, corr_3d, nn_proj1[..., 0], sampled_2d=sampled[:, :corr_2d.shape[1]],
                                              feat_3d_tail=last_flow_3d[:, :2], tail_scale=((w - 1, sw - 1), (h - 1, sh - 1)))
            else:
                with t.span("grid_sample"):  # corr fuser 3D (:376; utils via RPEFlow_core.py:107-108)
                    sampled = grid_sample_wrapper(torch.cat([corr_2d, self.flow_2d[level]], 1), xy1)
                    grid_sample_wrapper(ef_2d, xy1)
                with t.span("project_feat"):  # corr fuser 2D (:373)
                    flow_3d_to_2d = last_flow_3d[:, :2] * scale
                    project_feat_with_nn_corr(xy1, corr_2d, torch.cat([corr_3d, flow_3d_to_2d], 1), nn_proj1[..., 0])

            with t.span("flow_estimator_3d"):
                x_3d = [self.aligners[level](corr_3d), self.aligners[level](f1_3d), last_flow_3d, last_flow_feat_3d]
                if not getattr(self.flow_estimator_3d, "concatenates", False):
                    x_3d = torch.cat(x_3d, 1)
                flow_feat_3d = self.flow_estimator_3d(xyz1, x_3d, knn_1in1)
            with t.span("grid_sample"):  # decoder fusers (:394-395)
                sampled = grid_sample_wrapper(self.flow_feat_2d[level], xy1)
            with t.span("project_feat"):
                project_feat_with_nn_corr(xy1, self.flow_feat_2d[level], flow_feat_3d, nn_proj1[..., 0], **({"sampled_2d": sampled} if shares else {}))

            # conv_last_3d + the residual flow (RPEFlow_core.py:409-410): the 1x1 kernel with the residual in its epilogue, or a
            # library convolution and an add
            with t.span("flow_head_3d" if shares else "torch_glue"):
                if shares:
                    flows_3d.append(conv_module(self.flow_head_3d, flow_feat_3d, residual=last_flow_3d))
                else:
                    flows_3d.append(last_flow_3d + self.flow_head_3d(flow_feat_3d))
            flow_feats_3d.append(flow_feat_3d)

        flows_3d = flows_3d[::-1]
        with t.span("knn_interpolation"):  # final upsampling (:429-430)
            for i in range(len(flows_3d)):
                extra = {"knn_indices": interp_knn[i]} if i in interp_knn else {}
                flows_3d[i] = knn_interpolation(xyzs1[i + 1], flows_3d[i], xyzs1[i], **extra)
        return flows_3d[0], corr_2d
