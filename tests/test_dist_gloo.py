"""CPU, world_size 2 over gloo: the data-parallel scheme of SURVEY.md 8(e).
Rank k marches global rays [kR/2, (k+1)R/2) of the SAME global stream; the union of the shards must reproduce the
single-process batch exactly, the summed (all-reduced) gradients must equal the single-process gradient, and the
all-reduced counters must give every rank the same next rays_per_batch."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, q):
    sys.path[:0] = [HERE, os.path.join(os.path.dirname(HERE), "oracle"), os.path.join(os.path.dirname(HERE), "instant-ngp_amd")]
    import ngp_abi as A
    import oracle_py
    from common import OraModel, half_to_f32, host_meta, make_small_dataset, ptr
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ora = oracle_py.load()
    ora.ora_set_num_threads(2)
    imgs, xforms, meta = make_small_dataset(4, 32)
    M, X = host_meta(imgs, xforms, meta)
    grid = np.zeros(128 ** 3, np.float32)
    ora.ora_k_mark_untrained_density_grid(128 ** 3, ptr(grid), 4, M, X, 1)
    grid = np.where(grid >= 0, 0.05, grid).astype(np.float32)
    bf = np.zeros(128 ** 3, np.uint8)
    ora.ora_k_grid_to_bitfield(ptr(grid), 0, ptr(bf), C.c_float(0.01))
    n_rays, max_samples = 256, 1 << 18
    aabb = A.scene_aabb(1)
    rng = A.Pcg32(); ora.ora_pcg32_seed(C.byref(rng), C.c_uint64(1337), C.c_uint64(1))

    def k1(r, w):
        rc, nc = C.c_uint32(), C.c_uint32()
        ri = np.zeros(n_rays, np.uint32); rays = np.zeros((n_rays, 6), np.float32); ns = np.zeros((n_rays, 2), np.uint32); co = np.zeros((max_samples, 7), np.float32)
        ora.ora_k_generate_training_samples(n_rays, n_rays * r // w, n_rays * (r + 1) // w, aabb, max_samples, rng, C.byref(rc), C.byref(nc), ptr(ri), ptr(rays), ptr(ns), ptr(co),
                                            4, M, X, ptr(bf), 0, 1, C.c_float(0.0))
        return rc.value, nc.value, ri, ns, co

    rc, nc, ri, ns, co = k1(rank, world)
    # 1) shards are disjoint and their union is the single-process ray set with identical per-ray samples
    gathered = [None] * world
    dist.all_gather_object(gathered, (ri[:rc].tolist(), ns[:rc, 0].tolist()))
    rc1, nc1, ri1, ns1, co1 = k1(0, 1)
    union_idx = sum((g[0] for g in gathered), []); union_ns = sum((g[1] for g in gathered), [])
    assert sorted(union_idx) == sorted(ri1[:rc1].tolist()) and len(set(union_idx)) == len(union_idx)
    assert dict(zip(union_idx, union_ns)) == dict(zip(ri1[:rc1].tolist(), ns1[:rc1, 0].tolist()))
    cnt = torch.tensor([nc, rc], dtype=torch.int64); dist.all_reduce(cnt)
    assert cnt.tolist() == [nc1, rc1]
    # 2) gradients: each rank differentiates ITS samples with the loss normalised by the GLOBAL ray count; sum == full batch
    cfg = A.base_model_config(1, log2_hashmap_size=14)  # small table keeps the all-reduce cheap
    om = OraModel(ora, cfg)
    rs = np.random.default_rng(0)
    om.params_fp[om.n_mlp:] = rs.uniform(-1, 1, om.n - om.n_mlp).astype(np.float32); ora.ora_model_sync_half(om.h)
    def grads(coords, n):
        dl = np.zeros((n, 4), np.float16)
        # a deterministic per-sample output gradient that depends only on the sample itself (position hash) / global n_rays
        dl[:, :] = (np.sin(coords[:n, :3].sum(1) * 50.0)[:, None] * np.array([1.0, -0.5, 0.25, 2.0])[None, :] * (128.0 / n_rays)).astype(np.float16)
        om.training_step(np.ascontiguousarray(coords[:n]), dl.view(np.uint16))
        return half_to_f32(om.grads.copy()).astype(np.float64)
    g_local = torch.from_numpy(grads(co, nc))
    dist.all_reduce(g_local)
    g_full = grads(co1, nc1)
    err = np.linalg.norm(g_local.numpy() - g_full) / np.linalg.norm(g_full)
    assert err < 2e-2, err  # half-precision accumulation order differs between 1 and 2 shards
    # 3) every rank derives the same next rays_per_batch from the all-reduced counters (NerfCounters::update_after_training)
    B = 1 << 12
    compacted = torch.tensor([nc // 3], dtype=torch.int64); dist.all_reduce(compacted)
    nxt = min(((int(np.float32(n_rays) * np.float32(B * world) / np.float32(int(compacted)))) + 255) // 256 * 256, 1 << 18)
    allr = [None] * world
    dist.all_gather_object(allr, nxt)
    assert len(set(allr)) == 1
    if rank == 0:
        q.put("ok")
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_data_parallel_ray_shards_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(500)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) == "ok"
