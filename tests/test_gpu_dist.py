"""GPU: the multi-rank step of libngp_hip itself, not the oracle's.

gpurun hands out ONE GPU, and RCCL refuses two ranks on one device, so the world-size-2 test lets two processes share the GPU and
exchanges the two collectives' payloads through host memory over gloo -- everything else is the product path:
ngp_nerf_train_forward (K1 on this rank's slice of the global ray stream .. K4) -> all-reduce(sum) of the three counter words ->
ngp_nerf_train_backward (k_import_sync, controller, next K1, T1 / scatter / W) -> all-reduce(sum) of the fp16 gradients ->
ngp_nerf_train_finish (optimizer).  The RCCL calls themselves (ngp_comm_*, bucketed all-reduce inside ngp_nerf_train) run in the
second test with a communicator of one rank.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
B_GLOBAL = 1 << 18
RAYS0 = 256  # start where neither K1's sample cap nor K3's batch clamp drops rays (both are order dependent): ~500 samples per ray at initialisation
N_STEPS = 40


class _View:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def _setup(A, lib, rank, world, batch):
    from common import HipModel, host_meta, make_small_dataset
    imgs, xforms, meta = make_small_dataset(10, 80)
    M, X = host_meta(imgs, xforms, meta)
    cfg = A.base_model_config(1)
    hm = HipModel(lib, cfg)
    opts = A.default_nerf_options(1, target_batch_size=batch, rank=rank, world_size=world)
    t = C.c_void_p()
    A.check(lib, lib.ngp_nerf_create(hm.h, C.byref(opts), A.scene_aabb(1), C.byref(t)))
    pix = (C.c_void_p * len(imgs))(*[im.ctypes.data for im in imgs])
    A.check(lib, lib.ngp_nerf_set_dataset_host(t, len(imgs), M, X, pix))
    A.check(lib, lib.ngp_nerf_set_rays_per_batch(t, RAYS0))
    return hm, t, (imgs, M, X, pix)


def _scratch(A, lib, t, batch):
    ri, rays, ns, co, mo, cc, dl, cnt = (C.c_void_p() for _ in range(8))
    A.check(lib, lib.ngp_nerf_scratch_ptrs(t, C.byref(ri), C.byref(rays), C.byref(ns), C.byref(co), C.byref(mo), C.byref(cc), C.byref(dl), C.byref(cnt)))
    return dict(ray_indices=torch.as_tensor(_View(ri.value, 1 << 18, "<u4".replace("u", "i")), device="cuda"), numsteps=torch.as_tensor(_View(ns.value, 2 << 18, "<i4"), device="cuda"),
                coords_compacted=torch.as_tensor(_View(cc.value, batch * 7, "<f4"), device="cuda"), dloss=torch.as_tensor(_View(dl.value, batch * 4, "<i2"), device="cuda"),
                counters=torch.as_tensor(_View(cnt.value, 8, "<i4"), device="cuda"))  # TrainCounters: [4] = ray_counter (active rays of the step in flight)


def _rows(sc):
    """per global ray index: the bytes of its compacted coordinate / loss-gradient rows (valid rows only, before K4's padding)"""
    n_rays_active = int(sc["counters"].cpu()[4])
    ri = sc["ray_indices"].cpu().numpy().astype(np.uint32)[:n_rays_active]
    ns = sc["numsteps"].cpu().numpy().astype(np.uint32).reshape(-1, 2)[:n_rays_active]
    cc = sc["coords_compacted"].cpu().numpy().reshape(-1, 7); dl = sc["dloss"].cpu().numpy().reshape(-1, 4)
    out = {}
    for r, (k, b) in zip(ri, ns):
        if k:
            out[int(r)] = (cc[b:b + k].tobytes(), dl[b:b + k].tobytes())
    return out


def _worker(rank, world, port, q):
    sys.path[:0] = [HERE, os.path.join(os.path.dirname(HERE), "instant-ngp_amd")]
    import ngp_abi as A
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    lib = A.load_hip()
    hm, t, keep = _setup(A, lib, rank, world, B_GLOBAL // world)  # strong scaling: B / G samples per rank, the union is the 1-rank batch
    g = C.c_void_p(); lib.ngp_model_param_ptrs(hm.h, None, None, None, C.byref(g))
    grads = torch.as_tensor(_View(g.value, hm.n, "<f2"), device="cuda")
    cp = C.c_void_p(); lib.ngp_nerf_counter_ptrs(t, C.byref(cp))
    cnt = torch.as_tensor(_View(cp.value, 3, "<i4"), device="cuda")  # {marched, compacted, loss sum in units of 2^-24}
    sc = _scratch(A, lib, t, B_GLOBAL // world)
    ref = None
    if rank == 0:  # the single-rank trainer on the same GPU, same seed
        hm1, t1, keep1 = _setup(A, lib, 0, 1, B_GLOBAL)
        sc1 = _scratch(A, lib, t1, B_GLOBAL)
        g1 = C.c_void_p(); lib.ngp_model_param_ptrs(hm1.h, None, None, None, C.byref(g1))
        grads1 = torch.as_tensor(_View(g1.value, hm1.n, "<f2"), device="cuda")
        ref = {}
    losses = []
    for step in range(N_STEPS):
        A.check(lib, lib.ngp_nerf_train_prep(t, None))
        A.check(lib, lib.ngp_nerf_train_forward(t, None))
        torch.cuda.synchronize()
        c_host = cnt.cpu().to(torch.int64)
        rows = _rows(sc) if step == 0 else None
        dist.all_reduce(c_host)
        cnt.copy_(c_host.to(torch.int32).cuda())
        A.check(lib, lib.ngp_nerf_train_backward(t, None))
        torch.cuda.synchronize()
        g_local = grads.float().cpu()
        g_sum = g_local.clone(); dist.all_reduce(g_sum)
        grads.copy_(g_sum.half().cuda())
        A.check(lib, lib.ngp_nerf_train_finish(t, None))
        torch.cuda.synchronize()
        st = A.NerfStats(); A.check(lib, lib.ngp_nerf_get_stats(t, None, C.byref(st)))
        # (iii) every rank derives the same next rays_per_batch from the all-reduced counters
        rpb = [None] * world; dist.all_gather_object(rpb, int(st.rays_per_batch))
        assert len(set(rpb)) == 1, (step, rpb)
        losses.append(st.loss)
        if rank == 0:
            A.check(lib, lib.ngp_nerf_train_prep(t1, None))
            A.check(lib, lib.ngp_nerf_train_forward(t1, None))
            torch.cuda.synchronize()
            rows1 = _rows(sc1) if step == 0 else None
            A.check(lib, lib.ngp_nerf_train_backward(t1, None))
            torch.cuda.synchronize()
            g_one = grads1.float().cpu()
            A.check(lib, lib.ngp_nerf_train_finish(t1, None))
            st1 = A.NerfStats(); A.check(lib, lib.ngp_nerf_get_stats(t1, None, C.byref(st1)))
            ref[step] = dict(loss=st1.loss, rpb=int(st1.rays_per_batch), n_rays=int(st1.n_rays_last), measured=int(st1.measured_batch_size))
        if step == 0:
            # (i) step 0 starts from identical parameters: the union of the shards' rays is the single-rank ray set, and every ray's compacted
            #     coordinates and loss gradients are bit-identical (K3 normalises by the GLOBAL ray count)
            n_act = [None] * world; dist.all_gather_object(n_act, int(st.n_rays_last))
            mine = {k: v for k, v in rows.items()}
            allrows = [None] * world; dist.all_gather_object(allrows, mine)
            if rank == 0:
                union = {}
                for d in allrows:
                    assert not (set(d) & set(union)), "shards overlap"
                    union.update(d)
                one = rows1
                assert sum(n_act) == ref[0]["n_rays"] and set(union) == set(one), (n_act, ref[0], len(union), len(one))
                assert all(union[k] == one[k] for k in one), "per-ray compacted rows differ between 1 and 2 ranks"
                # (ii) the all-reduced gradient is the single-rank gradient up to K4's padding (each rank wraps ITS rows to B/G: not linear) and fp16 order
                a, b = g_sum.numpy().astype(np.float64), g_one.numpy().astype(np.float64)
                rel = float(np.linalg.norm(a - b) / np.linalg.norm(b))
                print(f"step 0: {len(one)} rays, summed-gradient rel-L2 vs single rank {rel:.3e}")
                assert rel < 0.2, rel   # measured 0.073 / 0.105 (255 rays at B = 2^17: nearly every row of a rank's half batch is padding, wrapped per rank)
    # (iv) replicated optimizer on identical reduced gradients: both ranks hold bit-identical parameters after N steps
    p = np.empty(hm.n, np.float32)
    A.check(lib, lib.ngp_model_get_params_host(hm.h, p.ctypes.data_as(C.c_void_p), C.c_uint64(p.size)))
    import hashlib
    digests = [None] * world; dist.all_gather_object(digests, hashlib.sha1(p.tobytes()).hexdigest())  # (hash() of bytes is salted per process)
    assert len(set(digests)) == 1, "ranks diverged"
    if rank == 0:
        # (v) the 2-rank run trains like the 1-rank run (same global batch, same ray stream); Testbed.loss is the loss of the UNION batch on
        #     every rank (the third all-reduced word).  Two trajectories of 40 noisy steps: same order of magnitude is all that can be asked here,
        #     the tight comparison is (vi)
        l2, l1 = losses[-1], ref[N_STEPS - 1]["loss"]
        print(f"loss after {N_STEPS} steps: 2 ranks {l2:.5f}, 1 rank {l1:.5f}; rays/batch {rpb[0]} vs {ref[N_STEPS - 1]['rpb']}")
        assert np.isfinite(l2) and l2 < 0.5 * losses[0] and l1 < 0.5 * ref[0]["loss"] and 0.6 < l2 / l1 < 1 / 0.6
        assert abs(rpb[0] - ref[N_STEPS - 1]["rpb"]) <= 0.15 * ref[N_STEPS - 1]["rpb"] + 256
    # (vi) steady state (n_in ~ B): every trainer takes ONE step from the single-rank trainer's state after N steps -- same parameters, occupancy
    #      grid, ray stream and rays per batch (slightly below the controller's value so that neither K1's sample cap nor K3's batch clamp, which
    #      drop order-dependent rays, binds).  The loss of the union batch must agree to fp32 summation order and the summed gradient to fp16
    #      rounding; K4's padding (each rank wraps ITS rows to B / G) is the only non-linear term and is measured separately.
    grid_n = 128 ** 3
    state = [None]
    if rank == 0:
        p1 = np.empty(hm1.n, np.float32)
        A.check(lib, lib.ngp_model_get_params_host(hm1.h, p1.ctypes.data_as(C.c_void_p), C.c_uint64(p1.size)))
        gp = C.c_void_p(); A.check(lib, lib.ngp_nerf_density_grid_ptrs(t1, C.byref(gp), None, None))
        grid1 = torch.as_tensor(_View(gp.value, grid_n, "<f4"), device="cuda").cpu().numpy().copy()
        r1, r1g = A.Pcg32(), A.Pcg32(); A.check(lib, lib.ngp_nerf_get_rng(t1, C.byref(r1), C.byref(r1g)))
        state[0] = dict(params=p1, grid=grid1, rng=(int(r1.state), int(r1.inc)), rays=int(ref[N_STEPS - 1]["rpb"] * 0.90) // 256 * 256)  # 10 % below the controller's value: the two ranks' shares of the samples differ by a few per cent
    dist.broadcast_object_list(state, 0)
    st0 = state[0]
    results = {}
    for zero_pad in (0, 1):
        lib.ngp_debug_set_flags(A.DBG_K4_ZERO_PADDING if zero_pad else 0)
        trainers = [(hm, t, grads, cnt, world)] + ([(hm1, t1, grads1, None, 1)] if rank == 0 else [])
        for (m_, t_, g_, c_, w_) in trainers:
            A.check(lib, lib.ngp_model_set_params_host(m_.h, st0["params"].ctypes.data_as(C.c_void_p), C.c_uint64(st0["params"].size)))
            A.check(lib, lib.ngp_nerf_set_density_grid_host(t_, None, st0["grid"].ctypes.data_as(C.c_void_p), C.c_uint64(grid_n)))
            rr = A.Pcg32(st0["rng"][0], st0["rng"][1]); A.check(lib, lib.ngp_nerf_set_rng(t_, C.byref(rr)))
            A.check(lib, lib.ngp_nerf_set_rays_per_batch(t_, st0["rays"]))
            A.check(lib, lib.ngp_nerf_train_forward(t_, None))
            torch.cuda.synchronize()
            if w_ > 1:
                c_host = c_.cpu().to(torch.int64); dist.all_reduce(c_host); c_.copy_(c_host.to(torch.int32).cuda())
            A.check(lib, lib.ngp_nerf_train_backward(t_, None))
            torch.cuda.synchronize()
            gl = g_.float().cpu()
            if w_ > 1:
                dist.all_reduce(gl)
                g_.copy_(gl.half().cuda())
            A.check(lib, lib.ngp_nerf_train_finish(t_, None))
            torch.cuda.synchronize()
            st_ = A.NerfStats(); A.check(lib, lib.ngp_nerf_get_stats(t_, None, C.byref(st_)))
            results[(zero_pad, w_)] = (gl.numpy().astype(np.float64), float(st_.loss), int(st_.measured_batch_size), int(st_.n_rays_last))
    lib.ngp_debug_set_flags(0)
    if rank == 0:
        for zero_pad in (0, 1):
            (g2, l2, m2, n2), (g1, l1, m1, n1) = results[(zero_pad, world)], results[(zero_pad, 1)]
            rel = float(np.linalg.norm(g2 - g1) / np.linalg.norm(g1))
            print(f"steady state, padding {'zeroed' if zero_pad else 'as in production'}: loss 2 ranks {l2:.6f} vs 1 rank {l1:.6f}; compacted per rank {m2} vs {m1} of {B_GLOBAL}; "
                  f"summed-gradient rel-L2 {rel:.3e}")
            assert m1 > 0.8 * B_GLOBAL and m1 < B_GLOBAL and m2 < B_GLOBAL // world, f"the check must run below the batch clamp ({m1}, {m2} of {B_GLOBAL})"
            assert abs(l2 - l1) <= 1e-3 * l1, (l2, l1)   # the loss of the union batch (Testbed.loss) on every rank: equal to six digits in every run so far
            # without the padding rows the step is linear in the set of rays: 2-rank sum == 1-rank gradient up to half rounding of the partial sums;
            # with them the difference is the wrap of each rank's first rows (a few per cent of the batch at n_in ~ 0.97 B)
            if zero_pad:
                assert rel < 1e-3, rel  # measured 2.6e-4: THE parity statement of the sharded step
            else:
                # The only other term is the padding: the single rank wraps its first (B - m1) rows, each of the two ranks its own first (B / 2 - m2) rows -- different rows, counted twice.
                # Per-sample gradients at a trained state are noise dominated (nearly uncorrelated), so a set of f N wrapped rows carries ~ sqrt(f) of the gradient's norm and the two
                # paddings differ by ~ sqrt(f1 + f2) of it (fully correlated gradients: f1 + f2).  Measured 0.038 - 0.181 at f = 0.05 - 0.11 over rounds 3 - 4 (a fixed 0.15 failed once).
                f1, f2 = 1.0 - m1 / B_GLOBAL, 1.0 - m2 / (B_GLOBAL // world)
                assert rel < 1.25 * np.sqrt(f1 + f2), (rel, f1, f2)
        q.put("ok")
    dist.barrier()
    lib.ngp_nerf_destroy(t)
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_multi_rank_step_world2_shared_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(800)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) == "ok"


def _sharded_worker(rank, world, port, q):
    """Two trainers per process on the same ray stream: one steps with the all-reduce + replicated-sweep step (rounds 2-4), one with the sharded step (reduce-scatter ->
    Adam on this rank's pieces -> all-gather -> foreign EMA).  The collectives run over gloo through host memory (one GPU is shared); everything else is the library."""
    sys.path[:0] = [HERE, os.path.join(os.path.dirname(HERE), "instant-ngp_amd")]
    import ngp_abi as A
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    lib = A.load_hip()
    lib.ngp_debug_set_flags(1048576)  # DBG_K3_TWO_PASS: deterministic compaction, so that the two trainers of a process see the same batch rows in the same order
    tr = {}
    for mode in ("allreduce", "sharded"):
        hm, t, keep = _setup(A, lib, rank, world, B_GLOBAL // world)
        g = C.c_void_p(); pp = C.c_void_p(); lib.ngp_model_param_ptrs(hm.h, None, C.byref(pp), None, C.byref(g))
        cp = C.c_void_p(); lib.ngp_nerf_counter_ptrs(t, C.byref(cp))
        tr[mode] = dict(hm=hm, t=t, keep=keep, grads=torch.as_tensor(_View(g.value, hm.n, "<f2"), device="cuda"), params=torch.as_tensor(_View(pp.value, hm.n, "<f2"), device="cuda"),
                        cnt=torch.as_tensor(_View(cp.value, 3, "<i4"), device="cuda"))
    sh = tr["sharded"]
    A.check(lib, lib.ngp_nerf_dp_set_sharded(sh["t"], 1))
    b = (C.c_uint64 * 2)(); e = (C.c_uint64 * 2)(); A.check(lib, lib.ngp_nerf_dp_layout(sh["t"], b, e))
    n_mlp = sh["hm"].n_mlp
    assert b[0] == n_mlp and e[0] == b[1] and e[1] == sh["hm"].n and all((e[k] - b[k]) % (4 * world) == 0 for k in range(2)), (list(b), list(e))
    pieces = [(int(b[k]) + (int(e[k]) - int(b[k])) // world * rank, (int(e[k]) - int(b[k])) // world) for k in range(2)]
    n_steps = 24
    for step in range(n_steps):
        for mode in ("allreduce", "sharded"):
            d = tr[mode]; t = d["t"]
            A.check(lib, lib.ngp_nerf_train_prep(t, None))
            A.check(lib, lib.ngp_nerf_train_forward(t, None))
            torch.cuda.synchronize()
            c_host = d["cnt"].cpu().to(torch.int64); dist.all_reduce(c_host); d["cnt"].copy_(c_host.to(torch.int32).cuda())
            A.check(lib, lib.ngp_nerf_train_backward(t, None))
            torch.cuda.synchronize()
            g_sum = d["grads"].float().cpu(); dist.all_reduce(g_sum); g_sum = g_sum.half()
            if mode == "allreduce":
                d["grads"].copy_(g_sum.cuda())
                A.check(lib, lib.ngp_nerf_train_finish(t, None))
            else:
                # reduce-scatter: only the MLP block and this rank's pieces receive the sums; the foreign pieces keep this rank's local (partial) gradients
                d["grads"][:n_mlp].copy_(g_sum[:n_mlp].cuda())
                for (o, n) in pieces:
                    d["grads"][o:o + n].copy_(g_sum[o:o + n].cuda())
                A.check(lib, lib.ngp_nerf_train_finish_sharded(t, None, 0))
                torch.cuda.synchronize()
                # all-gather of the pieces' new half parameters
                for k in range(2):
                    o, n = pieces[k]
                    mine = d["params"][o:o + n].cpu().view(torch.int32)   # (pieces are whole 4-half entries; gloo has no 16-bit integer type)
                    got = [torch.empty_like(mine) for _ in range(world)]
                    dist.all_gather(got, mine)
                    for r_ in range(world):
                        if r_ != rank:
                            o_r = int(b[k]) + n * r_
                            d["params"][o_r:o_r + n].copy_(got[r_].view(torch.float16).cuda())
                A.check(lib, lib.ngp_nerf_train_finish_sharded(t, None, 1))
            torch.cuda.synchronize()
        sa, ss = A.NerfStats(), A.NerfStats()
        A.check(lib, lib.ngp_nerf_get_stats(tr["allreduce"]["t"], None, C.byref(sa))); A.check(lib, lib.ngp_nerf_get_stats(sh["t"], None, C.byref(ss)))
        assert (sa.training_step, sa.rays_per_batch, sa.n_rays_last, sa.measured_batch_size) == (ss.training_step, ss.rays_per_batch, ss.n_rays_last, ss.measured_batch_size), (step, "counters")
    pa, ps = tr["allreduce"]["hm"], sh["hm"]
    for which in ("params", "inference"):
        x, y = pa.read(which, torch), ps.read(which, torch)
        n_diff = int((x != y).sum())
        assert n_diff == 0, f"rank {rank}: {n_diff} of {x.size} {which} halfs differ between the sharded and the all-reduce step after {n_steps} steps"
    ma, ms = pa.read("master", torch), ps.read("master", torch)
    own = np.zeros(ma.size, bool); own[:n_mlp] = True
    for (o, n) in pieces:
        own[o:o + n] = True
    assert np.array_equal(ma[own].view(np.uint32), ms[own].view(np.uint32)), "fp32 master parameters of the MLP / this rank's pieces"
    assert (ma[~own] != ms[~own]).any(), "the foreign pieces' fp32 state is NOT maintained by the sharded step (that is the saving)"
    g = sh["grads"].cpu().numpy().view(np.uint16)
    assert not g[n_mlp:].any(), "every grid gradient is cleared for the next step's scatter, foreign pieces included"
    # ... and the library says so: parameter read-backs / serialisation of the sharded trainer FAIL until the optimizer state has been gathered (they used to
    # return the stale foreign pieces silently), the all-reduce trainer's do not
    assert lib.ngp_nerf_dp_state_stale(sh["t"]) == 1 and lib.ngp_nerf_dp_state_stale(tr["allreduce"]["t"]) == 0
    buf = np.empty(ps.n, np.float32)
    assert lib.ngp_model_get_params_host(ps.h, buf.ctypes.data_as(C.c_void_p), C.c_uint64(buf.size)) != 0 and b"ngp_nerf_dp_gather_state" in lib.ngp_last_error()
    lib.ngp_model_serialized_size.restype = C.c_uint64
    blob = np.empty(int(lib.ngp_model_serialized_size(ps.h, 1)), np.uint8)
    assert lib.ngp_model_serialize_host(ps.h, blob.ctypes.data_as(C.c_void_p), C.c_uint64(blob.size), 1) != 0
    A.check(lib, lib.ngp_model_get_params_host(pa.h, buf.ctypes.data_as(C.c_void_p), C.c_uint64(buf.size)))
    A.check(lib, lib.ngp_nerf_dp_state_gathered(sh["t"]))   # (what a caller with its own collectives says after exchanging the state; here only the flag is exercised)
    A.check(lib, lib.ngp_model_get_params_host(ps.h, buf.ctypes.data_as(C.c_void_p), C.c_uint64(buf.size)))
    print(f"rank {rank}: sharded == all-reduce step, {n_steps} steps, loss {ss.loss:.6f}; pieces {pieces}")
    q.put("ok")
    dist.barrier()
    lib.ngp_nerf_destroy(tr["allreduce"]["t"]); lib.ngp_nerf_destroy(sh["t"])
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_sharded_step_equals_allreduce_step_world2_shared_gpu():
    """Round 5's data-parallel step -- reduce-scatter of the fp16 gradients, Adam on each rank's 1 / G piece of the table (+ the replicated MLP), all-gather of the new
    half parameters, EMA of the foreign pieces from the gathered parameters -- leaves BIT-IDENTICAL half parameters and inference (EMA) parameters to the all-reduce +
    replicated-sweep step, on both ranks, after 24 steps; the fp32 state of the foreign pieces is not maintained (checked: it differs) and all gradients are cleared."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(800)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) == "ok" and q.get(timeout=5) == "ok"


def test_grid_samples_drawn_ahead_are_the_updates_own(hip):
    """The occupancy-grid update's samples (generate_grid_samples_nerf_nonuniform twice + their sort, testbed_nerf.cu:2525-2557) are drawn on a side stream right after the PREVIOUS
    update, while the steps in between train: they depend on the grid rng, the EMA step and the grid as that update left it, on no parameter.  Asserted as what it must be: 340
    deterministic training steps (DBG_K3_TWO_PASS, see test_rccl_in_library_world1) with the samples drawn ahead vs inside the update (DBG2 bit 4): BIT-IDENTICAL density grid,
    bitfield, parameters and EMA parameters -- including an update whose samples ahead must be dropped (the dataset is set again in between: the grid may have been re-marked)."""
    import ngp_abi as A
    K3_TWO_PASS = 1048576
    GRID_NO_AHEAD = 4
    def run(flags2, reset_at):
        hip.ngp_debug_set_flags(K3_TWO_PASS); hip.ngp_debug_set_flags2(flags2)
        try:
            hm, t, keep = _setup(A, hip, 0, 1, 1 << 16)
            for k in range(17):
                A.check(hip, hip.ngp_nerf_train(t, None, 20))
                if k == reset_at:
                    imgs, M, X, pix = keep
                    A.check(hip, hip.ngp_nerf_set_dataset_host(t, len(imgs), M, X, pix))
            torch.cuda.synchronize()
            g, b, m = C.c_void_p(), C.c_void_p(), C.c_void_p()
            A.check(hip, hip.ngp_nerf_density_grid_ptrs(t, C.byref(g), C.byref(b), C.byref(m)))
            grid = torch.as_tensor(_View(g.value, 128 ** 3, "<i4"), device="cuda").cpu().numpy().copy()
            bits = torch.as_tensor(_View(b.value, 128 ** 3 // 8 // 4, "<i4"), device="cuda").cpu().numpy().copy()
            st = A.NerfStats(); A.check(hip, hip.ngp_nerf_get_stats(t, None, C.byref(st)))
            hip.ngp_nerf_grid_ahead_hits.restype = C.c_uint32
            out = dict(grid=grid, bits=bits, master=hm.read("master", torch), inference=hm.read("inference", torch), hits=int(hip.ngp_nerf_grid_ahead_hits(t)), step=st.training_step,
                       rays=st.rays_per_batch, batch=st.measured_batch_size)
            hip.ngp_nerf_destroy(t)
            return out
        finally:
            hip.ngp_debug_set_flags(0); hip.ngp_debug_set_flags2(0)
    ref = run(GRID_NO_AHEAD, 15)
    got = run(0, 15)
    assert ref["hits"] == 0 and ref["step"] == got["step"] == 340
    # updates every 16 steps from step 256 on; the first ones are drawn from step 272 on; the one after the dataset was set again (step 320) must not use what was drawn before it
    print(f"grid samples ahead: {got['hits']} updates found their samples ready; rays per batch {got['rays']} vs {ref['rays']}")
    assert 2 <= got["hits"] <= 4, got["hits"]
    assert (got["rays"], got["batch"]) == (ref["rays"], ref["batch"])
    for k in ("grid", "bits", "master", "inference"):
        a, b = ref[k], got[k]
        n_diff = int((a.view(np.uint32) != b.view(np.uint32)).sum()) if a.dtype.itemsize == 4 else int((a != b).sum())
        assert n_diff == 0, f"{k}: {n_diff} of {a.size} words differ between the samples drawn ahead and the samples drawn inside the update"


def test_rccl_in_library_world1(hip):
    """ngp_comm_unique_id / ngp_comm_init / the all-reduces inside ngp_nerf_train with a communicator of ONE rank: the collectives are identities, so training IS the plain
    single-rank run -- asserted as what an identity is: the same ray / sample counters on every step checked and BIT-IDENTICAL parameters after 30 steps.  (The data-parallel
    step takes the separate optimizer sweep where the plain step updates the hashed levels in k_grad_accumulate's epilogue: the two are bit-identical,
    tests/test_gpu_train.py::test_fused_optimizer_epilogue_is_the_separate_sweep.)  The one order-dependent piece of a training step, K3's span reservation by atomics (which
    rays the batch clamp drops, and the row order W sums over), is replaced by its deterministic two-pass variant for BOTH runs (ablation DBG_K3_TWO_PASS, slot-ordered
    compaction): every other kernel of the step is deterministic -- K1's prefix-sum spans, the lazy K2 per ray, exact fixed-point sums in the record lists of every level, W's
    fixed-order reduction tree, max-splat of the occupancy grid."""
    import ngp_abi as A
    K3_TWO_PASS = 1048576
    hip.ngp_debug_set_flags(K3_TWO_PASS)
    try:
        hm_a, t_a, keep_a = _setup(A, hip, 0, 1, 1 << 16)
        hm_b, t_b, keep_b = _setup(A, hip, 0, 1, 1 << 16)
        uid = (C.c_uint8 * 128)()
        A.check(hip, hip.ngp_comm_unique_id(uid))
        A.check(hip, hip.ngp_comm_init(t_b, 0, 1, uid))
        sa, sb = A.NerfStats(), A.NerfStats()
        for _ in range(3):
            A.check(hip, hip.ngp_nerf_train(t_a, None, 10))
            A.check(hip, hip.ngp_nerf_train(t_b, None, 10))
            torch.cuda.synchronize()
            A.check(hip, hip.ngp_nerf_get_stats(t_a, None, C.byref(sa))); A.check(hip, hip.ngp_nerf_get_stats(t_b, None, C.byref(sb)))
            assert (sa.training_step, sa.rays_per_batch, sa.n_rays_last, sa.measured_batch_size) == (sb.training_step, sb.rays_per_batch, sb.n_rays_last, sb.measured_batch_size), \
                [(s_.training_step, s_.rays_per_batch, s_.n_rays_last, s_.measured_batch_size) for s_ in (sa, sb)]
        assert sa.training_step == 30 and sa.measured_batch_size > 0
        pa, pb = hm_a.read("master", torch), hm_b.read("master", torch)
        n_diff = int((pa.view(np.uint32) != pb.view(np.uint32)).sum())
        print(f"rccl world-1 vs plain: loss {sb.loss:.8f} vs {sa.loss:.8f}; rays per batch {sb.rays_per_batch}; parameters that differ {n_diff} of {pa.size}")
        assert n_diff == 0, f"{n_diff} of {pa.size} master parameters differ between the one-rank communicator and the plain run (max |delta| {np.abs(pa - pb).max():.3e})"
        assert np.array_equal(hm_a.read("inference", torch), hm_b.read("inference", torch))   # EMA parameters
        # the loss: a float sum by atomics in the plain step, an integer sum in units of 2^-24 per ray under the communicator -> equal to rounding, not bit for bit
        assert np.isfinite(sb.loss) and abs(sa.loss - sb.loss) <= 1e-4 * sa.loss + 1e-7, (sa.loss, sb.loss)
    finally:
        hip.ngp_debug_set_flags(0)
    # the sharded step's structure with the same one-rank communicator kind (NGP_DP_SHARDED=1: reduce-scatter / all-gather of one rank are in-place identities; bucket A's
    # exchange runs on the communication stream behind its event, Adam runs piece by piece, the "foreign" EMA pass is empty): again bit-identical to the plain run
    hip.ngp_debug_set_flags(K3_TWO_PASS)
    os.environ["NGP_DP_SHARDED"] = "1"
    try:
        hm_c, t_c, keep_c = _setup(A, hip, 0, 1, 1 << 16)
        uid2 = (C.c_uint8 * 128)()
        A.check(hip, hip.ngp_comm_unique_id(uid2))
        A.check(hip, hip.ngp_comm_init(t_c, 0, 1, uid2))
        b2 = (C.c_uint64 * 2)(); e2 = (C.c_uint64 * 2)(); A.check(hip, hip.ngp_nerf_dp_layout(t_c, b2, e2))
        assert e2[1] == hm_c.n and b2[0] == hm_c.n_mlp and b2[1] == e2[0] and e2[0] > b2[0], "the sharded layout must be active"
        A.check(hip, hip.ngp_nerf_train(t_c, None, 30))
        torch.cuda.synchronize()
        sc_ = A.NerfStats(); A.check(hip, hip.ngp_nerf_get_stats(t_c, None, C.byref(sc_)))
        assert (sc_.training_step, sc_.rays_per_batch, sc_.measured_batch_size) == (sa.training_step, sa.rays_per_batch, sa.measured_batch_size)
        pc = hm_c.read("master", torch)
        n_diff = int((pa.view(np.uint32) != pc.view(np.uint32)).sum())
        print(f"rccl world-1 SHARDED step vs plain: loss {sc_.loss:.8f} vs {sa.loss:.8f}; parameters that differ {n_diff} of {pa.size}")
        assert n_diff == 0 and np.array_equal(hm_a.read("inference", torch), hm_c.read("inference", torch))
        A.check(hip, hip.ngp_nerf_dp_gather_state(t_c, None))   # (one rank: nothing to gather)
        A.check(hip, hip.ngp_comm_destroy(t_c)); hip.ngp_nerf_destroy(t_c)
    finally:
        os.environ.pop("NGP_DP_SHARDED", None); hip.ngp_debug_set_flags(0)
    A.check(hip, hip.ngp_allreduce_gradients(t_b, None)); A.check(hip, hip.ngp_allreduce_counters(t_b, None))
    # error-proportional pixel sampling under the communicator: the ranks' error maps are summed (fp32 all-reduce) before the CDFs are built
    opts = A.default_nerf_options(1, target_batch_size=1 << 16, rank=0, world_size=1, sample_image_proportional_to_error=1, sample_focal_plane_proportional_to_error=1)
    A.check(hip, hip.ngp_nerf_set_options(t_b, C.byref(opts)))
    A.check(hip, hip.ngp_nerf_set_error_map_interval(t_b, 4))
    A.check(hip, hip.ngp_nerf_train(t_b, None, 12))
    torch.cuda.synchronize()
    valid, nb = C.c_int(), C.c_uint32()
    A.check(hip, hip.ngp_nerf_error_map_ptrs(t_b, None, None, None, None, None, None, C.byref(valid), C.byref(nb), None))
    assert valid.value == 1 and nb.value == 9  # 4 -> 6 -> 9 after two updates
    A.check(hip, hip.ngp_nerf_get_stats(t_b, None, C.byref(sb)))
    assert np.isfinite(sb.loss) and sb.measured_batch_size > 0
    torch.cuda.synchronize()
    A.check(hip, hip.ngp_comm_destroy(t_b))
    hip.ngp_nerf_destroy(t_a); hip.ngp_nerf_destroy(t_b)


# ------------------------------------------------------------------------------------------------------------------------
# one process per GPU over RCCL inside the library -- the configuration the driver's scaling run uses (bench.py --gpus N)
# ------------------------------------------------------------------------------------------------------------------------
def _rccl_worker(rank, world, port, q):
    """ngp_nerf_train with an in-library communicator of `world` ranks, rank r on device r: every rank must derive the same ray counts, hold bit-identical parameters
    after the replicated optimizer steps (the all-reduced gradient is the same vector on every rank), and train (loss falls, stays finite)."""
    sys.path[:0] = [HERE, os.path.join(os.path.dirname(HERE), "instant-ngp_amd")]
    import hashlib
    import ngp_abi as A
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)                       # before the library creates its helper streams (they belong to the current device)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = A.load_hip()
    A.check(lib, lib.ngp_init())
    hm, t, keep = _setup(A, lib, rank, world, (1 << 16) // world)
    uid = [None]
    if rank == 0:
        buf = (C.c_uint8 * 128)(); A.check(lib, lib.ngp_comm_unique_id(buf)); uid[0] = bytes(buf)
    dist.broadcast_object_list(uid, 0)
    A.check(lib, lib.ngp_comm_init(t, rank, world, (C.c_uint8 * 128)(*uid[0])))
    losses = []
    for _ in range(4):
        A.check(lib, lib.ngp_nerf_train(t, None, 10))
        torch.cuda.synchronize()
        st = A.NerfStats(); A.check(lib, lib.ngp_nerf_get_stats(t, None, C.byref(st)))
        losses.append(float(st.loss))
        rpb = [None] * world; dist.all_gather_object(rpb, (int(st.rays_per_batch), int(st.training_step), float(st.loss)))
        assert len(set(rpb)) == 1, rpb               # same controller decision, same step, same (union-batch) loss on every rank
    # the half parameters every rank trains and renders with are kept identical by the step itself (sharded step: all-gather; all-reduce step: replicated sweep) ...
    ph = hm.read("params", torch); pi = hm.read("inference", torch)
    digests = [None] * world; dist.all_gather_object(digests, hashlib.sha1(ph.tobytes() + pi.tobytes()).hexdigest())
    assert len(set(digests)) == 1, "ranks diverged (half / inference parameters)"
    # ... the fp32 optimizer state of the foreign pieces only after the collective gather (sharded step, world > 1; a no-op otherwise)
    A.check(lib, lib.ngp_nerf_dp_gather_state(t, None))
    p = np.empty(hm.n, np.float32)
    A.check(lib, lib.ngp_model_get_params_host(hm.h, p.ctypes.data_as(C.c_void_p), C.c_uint64(p.size)))
    digests = [None] * world; dist.all_gather_object(digests, hashlib.sha1(p.tobytes()).hexdigest())
    assert len(set(digests)) == 1, "ranks diverged (fp32 master parameters after ngp_nerf_dp_gather_state)"
    assert np.isfinite(losses).all() and losses[-1] < 0.9 * losses[0], losses   # (losses[0] is already ten steps in)
    dist.barrier()
    torch.cuda.synchronize()
    A.check(lib, lib.ngp_comm_destroy(t))
    lib.ngp_nerf_destroy(t)
    dist.destroy_process_group()
    if rank == 0:
        q.put(("ok", losses))


def _run_rccl(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    tag, losses = q.get(timeout=5)
    print(f"in-library RCCL step, {world} rank(s) on {world} device(s): loss per 10 steps {losses}")
    assert tag == "ok"


@pytest.mark.timeout(700)
def test_rccl_step_one_process_per_device_world1():
    """the worker of the multi-device test below with a single rank (any box): same code path, communicator of one"""
    _run_rccl(1)


@pytest.mark.timeout(700)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the driver's 8-GPU box): RCCL across real devices")
def test_rccl_step_one_process_per_device_world2():
    _run_rccl(2)


@pytest.mark.timeout(900)
@pytest.mark.skipif(torch.cuda.device_count() < 8, reason="needs the 8-GPU node")
def test_rccl_step_one_process_per_device_world8():
    _run_rccl(8)
