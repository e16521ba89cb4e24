"""Stream ordering at the C-ABI (ABI v7, -m gpu): svo_wait_stream / svo_signal_stream order the context's kernels against
work on OTHER streams on the device, with no host synchronisation -- the contract hole behind round 4's red test (an output
tensor's zero fill on torch's stream racing the library's kernel on the context's stream).  Also: the validity of the
carried halo frame (SVO_CONTINUE_CARRY_FRAME) is tracked, not assumed."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tc():
    import torch
    assert torch.cuda.is_available()
    return torch


def test_handover_of_tensors_still_being_filled_on_another_stream(pkg, oracle, tc, small_seq):
    """Inputs produced and outputs zero-filled on a side stream that is kept busy: without an ordering call the library
    would read / be overwritten by unfinished work; with svo_wait_stream before and svo_signal_stream after, the results
    are the oracle's, and nothing ever synchronises with the host in between."""
    seq, frames = small_seq
    P1, P2 = seq.proj()
    imgs = [*frames[0], *frames[1]]
    h, w = imgs[0].shape
    c = pkg.Context(w, h, device=0, P1=P1, P2=P2)
    for s, im in enumerate(imgs):
        c.build_pyramid(s, im)
    kps = oracle.fast(imgs[0])
    pts = np.stack([kps["x"], kps["y"]], 1).astype(np.float32)
    res, _, _ = oracle.lk_track_step(oracle.make_params(P1, P2), *imgs, kps, np.eye(4), want_tracks=True)
    want_next, want_st = oracle.lk_track(imgs[0], imgs[1], pts)
    side = tc.cuda.Stream()
    busy = tc.empty(64 << 20, dtype=tc.float32, device="cuda")
    host_pts = tc.from_numpy(pts).pin_memory()
    for trial in range(4):
        with tc.cuda.stream(side):
            for _ in range(6):                               # several milliseconds of queued work in front of the hand-over
                busy.normal_()
            d_pts = host_pts.to("cuda", non_blocking=True)   # the input arrives on the side stream, late
            got, st = c.lk_track(0, 1, d_pts)                # binding: zero fill on `side`, svo_wait_stream(side), kernel, svo_signal_stream(side)
            out_h = got.to("cpu", non_blocking=True)         # consumer on the side stream, ordered by svo_signal_stream
            st_h = st.to("cpu", non_blocking=True)
        side.synchronize()
        assert st_h.numpy().tobytes() == want_st.tobytes(), trial
        assert out_h.numpy().tobytes() == want_next.tobytes(), trial
    # the raw entry points: a device chain whose output tensor is filled with garbage on the side stream right before
    T = tc.eye(4, dtype=tc.float64, device="cuda").reshape(1, 16).repeat(5, 1).contiguous()
    ok = tc.ones(5, dtype=tc.int32, device="cuda")
    with tc.cuda.stream(side):
        for _ in range(6):
            busy.normal_()
        out = tc.full((5, 16), 7.0, dtype=tc.float64, device="cuda")
        c.wait_stream(side.cuda_stream)
        c._check(c.lib.svo_chain_relative(c.h, C.c_void_p(T.data_ptr()), C.c_void_p(ok.data_ptr()), 5, None,
                                          C.c_void_p(out.data_ptr()), pkg.MEM_DEVICE))
        c.signal_stream(side.cuda_stream)
        back = out.to("cpu", non_blocking=True)
    side.synchronize()
    assert np.array_equal(back.numpy(), np.tile(np.eye(4).reshape(1, 16), (5, 1)))
    c.close()


def test_signal_stream_covers_the_overlapped_pose_stage(pkg, tc, synth):
    """Overlap mode ends a batch on the context's side stream: svo_signal_stream must hand THAT to the consumer stream."""
    seq = synth.StereoSequence(width=416, height=128, n_frames=6, seed=5, device=tc.device("cuda", 0))
    L = tc.stack([seq.render(t)[0] for t in range(6)])
    R = tc.stack([seq.render(t)[1] for t in range(6)])
    P1, P2 = seq.proj()
    c = pkg.Context(416, 128, device=0, P1=P1, P2=P2, max_batch=5)
    want = c.track_batch(L, R)
    c.set_overlap(True)
    consumer = tc.cuda.Stream()
    buf = tc.zeros((5, pkg.STEP_DTYPE.itemsize), dtype=tc.uint8, device="cuda")
    tc.cuda.synchronize()
    c.track_batch(L, R, results=buf)
    c.signal_stream(consumer.cuda_stream)
    with tc.cuda.stream(consumer):
        host = buf.to("cpu", non_blocking=True)
    consumer.synchronize()
    got = np.frombuffer(host.numpy().tobytes(), dtype=pkg.STEP_DTYPE)
    assert got["pose"].tobytes() == want["pose"].tobytes() and int(got["ok"].sum()) == int(want["ok"].sum()) >= 4
    c.close()


def test_carry_frame_needs_a_valid_previous_async_batch(pkg, synth):
    """SVO_CONTINUE_CARRY_FRAME is refused -- not silently served from stale frame slots -- when no async batch left its last
    frame behind, when it comes without SVO_CONTINUE_CHAIN, and after svo_add_frame or a synchronous batch wrote the slots."""
    seq = synth.StereoSequence(width=416, height=128, n_frames=5, seed=5)
    fr = [tuple(x.numpy() for x in seq.render(t)) for t in range(5)]
    P1, P2 = seq.proj()
    c = pkg.Context(416, 128, device=0, P1=P1, P2=P2, max_batch=2)
    hl, hr = c.host_frames(3), c.host_frames(3)

    def put(buf, a):
        for k in range(3):
            hl[k, :, :416], hr[k, :, :416] = fr[a + k]
        c.upload_frames(buf, hl, hr)
        c.wait_upload(buf)

    put(0, 0)
    with pytest.raises(pkg.SvoError):                        # nothing to carry yet
        c.track_uploaded_async(0, 3, continue_chain=True, carry_frame=True)
    c.track_uploaded_async(0, 3)
    r0 = c.collect_results(2)
    put(1, 2)
    with pytest.raises(pkg.SvoError):                        # CARRY without CHAIN
        c._check(c.lib.svo_track_uploaded_async(c.h, 1, 3, None, 2))
    c.track_uploaded_async(1, 3, continue_chain=True, carry_frame=True)      # the valid case
    r1 = c.collect_results(2)
    assert int(r0["ok"].sum()) == 2 and int(r1["ok"].sum()) == 2
    # a buffer uploaded FROM SLOT 1 (svo_upload_frames_at: frame 0 is to be carried on the device) holds stale pixels in slot 0:
    # only a launch that carries the previous batch's last frame may read it -- the synchronous entry and a chained launch
    # without CARRY are refused, and so is a first_slot beyond 1
    c.upload_frames(0, hl[1:3], hr[1:3], first_slot=1)
    c.wait_upload(0)
    with pytest.raises(pkg.SvoError):
        c.track_uploaded(0, 3)
    with pytest.raises(pkg.SvoError):
        c.track_uploaded_async(0, 3, continue_chain=True)
    with pytest.raises(pkg.SvoError):
        c.upload_frames(0, hl[2:3], hr[2:3], first_slot=2)
    c.track_uploaded_async(0, 3, continue_chain=True, carry_frame=True)      # ... and the launch it is meant for is accepted
    r2 = c.collect_results(2)
    assert int(r2["ok"].sum()) == 2
    c.add_frame(*fr[0])                                      # the online ring overwrites frame slots 0 / 1
    put(0, 2)
    with pytest.raises(pkg.SvoError):
        c.track_uploaded_async(0, 3, continue_chain=True, carry_frame=True)
    c.host_free(hl); c.host_free(hr)
    c.close()


@pytest.mark.parametrize("mode", ["orb", "lk"])
def test_track_batch_orders_frames_and_results_against_torchs_stream(pkg, tc, synth, mode):
    """Context.track_batch with device frames + device results on a context that runs on its OWN stream: the frames arrive
    late on a busy torch stream (non-blocking upload behind queued work), the result buffer's zero fill is a kernel on that
    stream, and right after the call the frames are dropped and their blocks refilled with noise.  ORB mode reads level 0 in
    place until the end of the front end, so without ordering on both sides the keypoints are garbage.  The binding orders
    torch's stream before the launch and the kernels that read the frames before torch's next operation
    (svo_signal_stream_inputs, ABI v9: two event operations each, the pose-stage overlap is kept)."""
    seq = synth.StereoSequence(width=416, height=128, n_frames=6, seed=5)
    fr = [seq.render(t) for t in range(6)]
    hostL = tc.stack([f[0] for f in fr]).pin_memory()
    hostR = tc.stack([f[1] for f in fr]).pin_memory()
    P1, P2 = seq.proj()
    kw = dict(track_mode=pkg.MODE_ORB, min_move2=0.05 ** 2, max_move2=100.0, orb_nlevels=3, orb_nfeatures=400) if mode == "orb" else {}
    c = pkg.Context(416, 128, device=0, P1=P1, P2=P2, max_batch=5, **kw)
    want = c.track_batch(hostL.cuda(), hostR.cuda())
    assert int(want["ok"].sum()) >= 4
    c.set_overlap(True)
    side = tc.cuda.Stream()
    busy = tc.empty(64 << 20, dtype=tc.float32, device="cuda")
    for trial in range(3):
        with tc.cuda.stream(side):
            for _ in range(6):
                busy.normal_()
            L = hostL.to("cuda", non_blocking=True)
            R = hostR.to("cuda", non_blocking=True)
            buf = tc.zeros((5, pkg.STEP_DTYPE.itemsize), dtype=tc.uint8, device="cuda")
            c.track_batch(L, R, results=buf)
            del L, R                                       # the caching allocator hands these blocks to the next tensors
            junk = [tc.randint(0, 255, hostL.shape, dtype=tc.uint8, device="cuda") for _ in range(2)]
            c.signal_stream(side.cuda_stream)              # the records: complete once the side-stream pose stage is (the caller's call)
            host = buf.to("cpu", non_blocking=True)
        side.synchronize()
        got = np.frombuffer(host.numpy().tobytes(), dtype=pkg.STEP_DTYPE)
        for k in ("ok", "n_prev_kps", "n_cur_kps", "n_tracked", "n_inliers"):
            assert got[k].tolist() == want[k].tolist(), (trial, k)
        assert got["pose"].tobytes() == want["pose"].tobytes(), trial
        del junk
    c.close()
