"""The reference's own three test programs, at the sizes they publish, as pytest cases.

models/csrc/correlation/correlation_test.cpp, k_nearest_neighbor/k_nearest_neighbor_test.cpp and
furthest_point_sampling/furthest_point_sampling_test.cpp are the only known-answer recipes the reference holds for this path
(BASELINE.md section 1): each builds seeded uniform inputs, runs the extension beside a naive PyTorch restatement and prints
a verdict and two wall-clock times.  Here the extension's seat is taken by rpeflow_amd.csrc (through the C ABI), the naive
side is the same PyTorch recipe / the CPU oracle (which restates the very recipe: matmul + topk, the argmax loop), the
verdicts are assertions -- stricter than the reference's where it only prints a count -- and the kernel times are printed
beside them (pytest -s), because these are the only workloads the reference itself times.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu

import rpeflow_amd.csrc as ops  # noqa: E402

DEV = "cuda:0"


def timed(fn, warmup=2, reps=3):
    """Median device time of fn() in ms (events on the current stream, which is the one the C ABI launches on)."""
    for _ in range(warmup):
        fn()
    times = []
    for _ in range(reps):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        out = fn()
        t1.record()
        torch.cuda.synchronize()
        times.append(t0.elapsed_time(t1))
    return out, sorted(times)[len(times) // 2]


def test_correlation_recipe_b32_c128_144x240_forward_and_both_gradients():
    """correlation_test.cpp:45-49 (B = 32, C = 128, 144 x 240, md = 4, rand in [0,1)), :27-42 (the naive implementation: 81
    slices of the zero-padded second input, mean over channels, cat; gradients through autograd), :82-89 (mean |diff| < 1e-6
    for the output and both gradients)."""
    B, C, H, W, md = 32, 128, 144, 240, 4
    torch.manual_seed(0)
    in1 = torch.rand(B, C, H, W, device=DEV, requires_grad=True)
    in2 = torch.rand(B, C, H, W, device=DEV, requires_grad=True)
    grad_out = torch.rand(B, (2 * md + 1) ** 2, H, W, device=DEV)

    padded = torch.nn.functional.pad(in2, (md, md, md, md))
    naive = torch.cat([(in1 * padded[:, :, i:i + H, j:j + W]).mean(1, keepdim=True)
                       for i in range(2 * md + 1) for j in range(2 * md + 1)], 1)
    naive.backward(grad_out)
    naive_g1, naive_g2 = in1.grad.clone(), in2.grad.clone()
    in1.grad = in2.grad = None

    out = ops.correlation2d(in1, in2, md)
    out.backward(grad_out)
    assert out.shape == (B, 81, H, W)
    d_out = (out.detach() - naive.detach()).abs()
    d_g1, d_g2 = (in1.grad - naive_g1).abs(), (in2.grad - naive_g2).abs()
    # the reference's three verdicts ...
    assert d_out.mean().item() < 1e-6 and d_g1.mean().item() < 1e-6 and d_g2.mean().item() < 1e-6
    # ... and a worst-element bound it does not have
    assert d_out.max().item() < 5e-6 and d_g1.max().item() < 5e-6 and d_g2.max().item() < 5e-6

    a, b = in1.detach(), in2.detach()
    _, fwd_ms = timed(lambda: ops.correlation2d(a, b, md))

    def both():
        in1.grad = in2.grad = None
        ops.correlation2d(in1, in2, md).backward(grad_out)
    _, both_ms = timed(both)
    print("\ncorrelation_test.cpp recipe (32x128x144x240, md 4): forward %.3f ms, forward + backward %.3f ms; mean |diff| %.2e / %.2e / %.2e"
          % (fwd_ms, both_ms, d_out.mean().item(), d_g1.mean().item(), d_g2.mean().item()))


def test_knn_recipe_b8_8192x8192_k16_has_zero_mismatches():
    """k_nearest_neighbor_test.cpp:25-37 (B = 8, 8192 input and 8192 query points, k = 16, rand in [0,1), seed 0), :16-22 (the
    matmul + topk restatement), :61-63 (the reference PRINTS how many of the 1 048 576 indices differ; here the count must be
    zero, in all eight batches, against the oracle's restatement of the CPU matmul + topk -- order of equal distances included)."""
    B, M, Q, k = 8, 8192, 8192, 16
    torch.manual_seed(0)
    inp, qry = torch.rand(B, M, 3), torch.rand(B, Q, 3)
    want = O.k_nearest_neighbor(inp.numpy(), qry.numpy(), k)
    d_inp, d_qry = inp.to(DEV), qry.to(DEV)
    got, ms = timed(lambda: ops.k_nearest_neighbor(d_inp, d_qry, k))
    assert got.dtype == torch.int64 and got.shape == (B, Q, k)
    mismatched = int((got.cpu().numpy() != want).sum())
    print("\nk_nearest_neighbor_test.cpp recipe (8 x 8192 x 8192, k 16): %.3f ms; %d of %d elements are mismatched" % (ms, mismatched, B * Q * k))
    assert mismatched == 0


def test_fps_recipe_b64_4096_to_1024_equals_the_argmax_loop():
    """furthest_point_sampling_test.cpp:34-36 (B = 64, 4096 points, 1024 samples, rand in [0,1), seed 0), :16-31 (the running
    minimum + argmax loop), :63 (torch::equal)."""
    B, N, S = 64, 4096, 1024
    torch.manual_seed(0)
    xyz = torch.rand(B, N, 3)
    want = O.furthest_point_sampling(xyz.numpy(), S)
    d_xyz = xyz.to(DEV)
    got, ms = timed(lambda: ops.furthest_point_sampling(d_xyz, S))
    print("\nfurthest_point_sampling_test.cpp recipe (64 x 4096 -> 1024): %.3f ms (%.3f us per dependent sample)" % (ms, ms * 1e3 / S))
    assert np.array_equal(got.cpu().numpy(), want)
