"""Stream-ordered interop with PyTorch (what bench.py relies on): kernels issued on torch's current stream see
tensors written just before and are seen by torch ops (and RCCL collectives) issued right after."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("use_side_stream", [False, True])
def test_device_pointers_on_torch_stream(oracle, use_side_stream):
    import torch
    import panovlm_amd as pv
    rng = np.random.default_rng(9)
    F, P = 12, 60
    aa, t = synth.random_poses(rng, F)
    ref, nei = synth.random_pairs(rng, F, P)
    rows, off = synth.random_resset(rng, 1, aa, t, ref, nei, rng.integers(200, 3000, size=P))
    ctx = pv.Context(0)
    rs = pv.ResidualSet.upload(ctx, 1, rows, off, ref, nei, flags=1)
    up = sorted({(min(a, b), max(a, b)) for a, b in zip(ref.tolist(), nei.tolist())})
    neq = pv.NormalEq(ctx, F, [u[0] for u in up], [u[1] for u in up])
    ctx.set_poses(aa, t)
    expect = neq.accumulate(rs, 1, 0.035)            # host-pointer path on the ctx's own stream
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev) if use_side_stream else torch.cuda.current_stream(dev)
    with torch.cuda.stream(side):
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        d_aa = torch.zeros((F, 3), dtype=torch.float64, device=dev)
        d_t = torch.zeros((F, 3), dtype=torch.float64, device=dev)
        packed = torch.full((neq.size,), float("nan"), dtype=torch.float64, device=dev)
        for _ in range(3):
            # torch writes the parameters, we consume them immediately, torch consumes our output immediately
            d_aa.copy_(torch.from_numpy(aa), non_blocking=False); d_t.copy_(torch.from_numpy(t), non_blocking=False)
            d_aa.mul_(1.0); d_t.add_(0.0)
            ctx.set_poses_dev(F, d_aa.data_ptr(), d_t.data_ptr())
            neq.accumulate_dev(rs, packed.data_ptr(), 1, 0.035, zero_first=True)
            doubled = packed * 2.0
        got = doubled.cpu().numpy() / 2.0
    ctx.use_own_stream()
    assert np.allclose(got, expect, rtol=1e-12, atol=1e-12 * np.abs(expect).max())
    r = torch.empty(rs.n, dtype=torch.float64, device=dev); J = torch.empty((rs.n, 12), dtype=torch.float64, device=dev)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    rs.eval_dev(r.data_ptr(), J.data_ptr())
    rh, Jh = r.cpu().numpy(), J.cpu().numpy()
    ctx.use_own_stream()
    r2, J2 = rs.eval()
    assert np.array_equal(rh, r2) and np.array_equal(Jh, J2)
    rs.close(); ctx.close()


def test_rccl_allreduce_through_the_abi_single_rank():
    """pvlm_comm_* / pvlm_allreduce_sum_f64 (the C++ hosts' exchange) with world_size 1: RCCL loads lazily, the
    communicator initialises and the in-place all-reduce of a packed buffer is the identity."""
    import torch
    import panovlm_amd as pv
    ctx = pv.Context(0)
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    comm = pv.Comm(ctx, 1, 0)
    assert len(comm.unique_id) == 128
    x = torch.arange(1000, dtype=torch.float64, device=dev) * 0.5
    ref = x.clone()
    comm.allreduce_sum_f64(x.data_ptr(), x.numel())
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    comm.close(); ctx.use_own_stream(); ctx.close()


def test_device_resident_panorama_maps_equal_the_host_pointer_entry_points():
    """pvlm_cam_to_image_f32_dev / pvlm_image_to_cam_f32_dev (four points per lane, 16-byte accesses + a scalar tail) against the
    host-pointer entry points (one point per lane): same floats, for sizes with every tail length and for an unaligned view."""
    import torch
    import panovlm_amd as pv
    ctx = pv.Context(0)
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    rows, cols = 2880, 5760
    g = torch.Generator(device="cpu"); g.manual_seed(3)
    for n in (1, 3, 4, 5, 1000, 1001, 1002, 1003, 70001):
        cam = (torch.randn((n + 1, 3), generator=g, dtype=torch.float32) * 3)
        px = (torch.rand((n + 1, 2), generator=g, dtype=torch.float32) * torch.tensor([cols, rows], dtype=torch.float32))
        for off in (0, 1):                         # off = 1: the device view starts 12 / 8 bytes into the allocation -> scalar path
            c_d = cam.to(dev)[off:off + n].contiguous() if off == 0 else cam.to(dev)[off:off + n]
            p_d = px.to(dev)[off:off + n].contiguous() if off == 0 else px.to(dev)[off:off + n]
            assert c_d.is_contiguous() and p_d.is_contiguous()
            o_px = torch.empty((n, 2), device=dev, dtype=torch.float32); o_cam = torch.empty((n, 3), device=dev, dtype=torch.float32)
            assert c_d.dtype == p_d.dtype == torch.float32
            ctx.cam_to_image_f32_dev(rows, cols, n, c_d.data_ptr(), o_px.data_ptr())
            ctx.image_to_cam_f32_dev(rows, cols, n, p_d.data_ptr(), 2.5, o_cam.data_ptr())
            torch.cuda.synchronize()
            want_px = ctx.cam_to_image(rows, cols, c_d.cpu().numpy())
            want_cam = ctx.image_to_cam(rows, cols, p_d.cpu().numpy(), 2.5)
            assert np.array_equal(o_px.cpu().numpy(), want_px, equal_nan=True), (n, off)
            assert np.array_equal(o_cam.cpu().numpy(), want_cam), (n, off)
    ctx.use_own_stream(); ctx.close()
