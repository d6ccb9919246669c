"""One rank of an N-rank run whose ranks SHARE a GPU: the whole trc_group_* / grouped SPPM path with the collectives
supplied through trc_group_set_collectives (host-staged, gloo between the processes) instead of RCCL, which refuses two
ranks on one device.  Started by tests/test_gpu_shared_gpu_ranks.py; not a test module itself.

env: RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT, TRC_ROOT, TRC_OUT (directory),
     TRC_CASE = small | sppm_small | ragged | config4 | config5 | samples2 | samples3 | samples4 (sample sharding at 1080p on BASELINE configs 2 / 3 / 4)
"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.environ["TRC_ROOT"])
import torch.distributed as dist  # noqa: E402

from tracer_amd import abi, host  # noqa: E402
from tracer_amd.device import Tracer  # noqa: E402
from tracer_amd.gloo_collectives import GlooCollectives  # noqa: E402


def owner_mask(W, H, world, rank):
    ty, tx = np.mgrid[0:H, 0:W] // abi.TRC_TILE
    return ((tx + ty) % world) == rank


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).tobytes()).hexdigest()


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    case, outdir = os.environ["TRC_CASE"], os.environ["TRC_OUT"]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    coll = GlooCollectives()
    out = {}
    t = Tracer(0)                                   # every rank on the same GPU
    if case == "small":
        W, H, spp = 320, 192, 6
        scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
        t.upload_scene(scene.view); t.set_camera(host.prepare_camera(W, H)); t.set_environment((0.0, 0.0, 0.0)); t.resize(W, H)
        t.set_collectives(coll, world, rank)
        # path-traced frame: synchronous compose, then three pipelined steps in flight
        t.clear_accum(); t.seed(31); t.render(spp=spp, tile_rank=rank, tile_nranks=world); t.group_reduce_accum(0)
        if rank == 0:
            out["sync"] = t.download_accum()
        for step in range(3):
            t.clear_accum(); t.seed(40 + step); t.render(spp=spp, tile_rank=rank, tile_nranks=world)
            t.group_reduce_accum_async(0)
            if rank == 0:
                out[f"async{step}"] = t.download_composed()
        # sample sharding (tracer_abi.h): every rank the whole frame with 16 / N samples from ITS seed; all-to-all of the
        # pixel slices, rank-ordered fold, gather to the root -- then the same compose delivered to every rank
        t.synchronize(); t.clear_accum(); t.seed(abi.shard_seed(100, rank)); t.render(spp=16 // world)
        mine = t.download_accum()
        t.group_compose_samples(0)
        if rank == 0:
            out["samples"] = t.download_composed()
        else:                                    # the composed frame exists on the root only: the others are told so, not handed stale slices
            try:
                t.download_composed()
                out["nonroot_refused"] = np.array(False)
            except Exception as e:
                out["nonroot_refused"] = np.array(getattr(e, "status", None) == abi.ERR_NO_FRAME)
        out["untouched"] = np.array(np.array_equal(mine.view(np.uint32), t.download_accum().view(np.uint32)))   # a progressive host renders on
        t.group_allreduce_mean_accum()
        out["mean"] = t.download_accum()
        # seeds x tiles: S = N / 2 seeds, each seed's frame tiled over 2 ranks; pipelined, two steps in flight
        if world >= 4:
            S, T = world // 2, 2
            for step in range(2):
                t.clear_accum(); t.seed(abi.shard_seed(200 + step, rank // T)); t.render(spp=16 // S, tile_rank=rank % T, tile_nranks=T)
                t.group_compose_samples_async(0, S)
                if rank == 0:
                    out[f"hybrid{step}"] = t.download_composed()
        # a progressive host: 4 samples, compose, 4 more ON TOP (frame0 continuing), compose -- pipelined, never cleared: the
        # pipelined compose works on a snapshot and leaves the accumulator to the renderer
        t.synchronize(); t.clear_accum(); t.seed(abi.shard_seed(300, rank))
        t.render(spp=4); t.group_compose_samples_async(0)
        t.render(spp=4, frame0=4); t.group_compose_samples_async(0)
        if rank == 0:
            out["prog8"] = t.download_composed()
        t.render(spp=2, frame0=8); t.group_compose_samples_async(0)
        if rank == 0:
            out["prog10"] = t.download_composed()
        # SPPM: bound keys all-reduced, photon records all-gathered, frame composed
        t.synchronize(); t.clear_accum(); t.seed(8); t.sppm_init(9); t.sppm_frames(3)
        cam, pho, mark, count, cx = t.sppm_download()
        t.group_reduce_accum(0)
        own = owner_mask(W, H, world, rank).ravel()
        out["cam_own"] = cam[own].view(np.uint8)
        out["pho"] = pho.view(np.uint8); out["mark"] = mark; out["count"] = count
        out["total"] = np.float32(cx.totalPhotonSum); out["box"] = np.array([cx.photonBox.mini.x, cx.photonBox.mini.y, cx.photonBox.mini.z,
                                                                              cx.photonBox.maxi.x, cx.photonBox.maxi.y, cx.photonBox.maxi.z], np.float32)
        if rank == 0:
            out["sppm"] = t.download_accum()
        out["calls"] = np.array([coll.calls[k] for k in ("reduce", "allreduce", "allgather", "alltoall", "gather")])
    elif case == "sppm_small":
        # a canvas of 3 x 2 tiles under more ranks than tiles, and rank counts that do not divide the photons: records, grids and frame
        W, H, frames = 40, 24, 4
        scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
        t.upload_scene(scene.view); t.set_camera(host.prepare_camera(W, H)); t.set_environment((0.0, 0.0, 0.0)); t.resize(W, H)
        t.set_collectives(coll, world, rank)
        t.clear_accum(); t.seed(8); t.sppm_init(9); t.sppm_frames(frames)
        cam, pho, mark, count, cx = t.sppm_download()
        t.group_reduce_accum(0)
        own = owner_mask(W, H, world, rank).ravel()
        out["cam_own"] = cam[own].view(np.uint8)
        out["pho"] = pho.view(np.uint8); out["mark"] = mark; out["count"] = count
        out["total"] = np.float32(cx.totalPhotonSum); out["radius"] = np.float32(cx.photonInitialRadius)
        if rank == 0:
            out["sppm"] = t.download_accum()
    elif case == "ragged":
        # a frame whose pixel count the ranks do not divide (97 x 61 = 5917: the last pixel slice is short), and a table that
        # cannot compose sample shards
        W, H = 97, 61
        scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
        t.upload_scene(scene.view); t.set_camera(host.prepare_camera(W, H)); t.set_environment((0.1, 0.2, 0.3)); t.resize(W, H)
        t.set_collectives(coll, world, rank)
        t.clear_accum(); t.seed(abi.shard_seed(5, rank)); t.render(spp=16 // world)
        t.group_compose_samples(0)
        if rank == 0:
            out["sync"] = t.download_composed()
        t.group_compose_samples_async(0, world)
        if rank == 0:
            out["async"] = t.download_composed()
        t.synchronize(); t.clear_accum(); t.seed(abi.shard_seed(5, rank)); t.render(spp=16 // world)
        t.group_allreduce_mean_accum()
        out["mean"] = t.download_accum()
        # the same ranks with a table that has no alltoall / gather: tiles still compose, sample shards say so
        import ctypes as C
        from tracer_amd import device
        coll.table.alltoall = C.cast(None, type(coll.table.alltoall))
        t.set_collectives(coll, world, rank)
        try:
            t.group_compose_samples(0)
            out["refused"] = np.array(0)
        except device.TracerError as e:
            out["refused"] = np.array(e.status)
        t.clear_accum(); t.seed(6); t.render(spp=2, tile_rank=rank, tile_nranks=world); t.group_reduce_accum(0)
        if rank == 0:
            out["tiles"] = t.download_accum()
    elif case in ("samples2", "samples3", "samples4"):
        # the split that scales, at the named size: every rank the WHOLE 1920x1080 frame, 64 / N samples from seed
        # trc_shard_seed(seed, rank), composed by trc_group_compose_samples_async (the bench's --scaling samples step)
        W, H, total = 1920, 1080, 64
        integ = abi.INTEGRATOR_PATH
        if case == "samples2":
            scene, seed = host.HostScene(abi.SCENE_CORNELL_SPHERES), 0x5EED0000
        elif case == "samples3":                            # BASELINE config 3's scene and integrator: coatball.obj, traceMIS
            scene, seed, integ = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.golden("coatball")), 0x5EED0003, abi.INTEGRATOR_MIS
        else:
            scene, seed = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.golden("teapot").replicate(8, 80.0)), 0x5EED0004
        t.upload_scene(scene.view); t.set_camera(host.prepare_camera(W, H)); t.set_environment((0.0, 0.0, 0.0)); t.resize(W, H)
        t.set_collectives(coll, world, rank)
        for launch in range(2):                              # the second launch runs in adaptive order, on the other accumulator
            t.reset_stats()
            t.clear_accum(); t.seed(abi.shard_seed(seed, rank)); t.render(spp=total // world, integrator=integ)
            t.group_compose_samples_async(0, world)
        st = t.stats()
        out["rays"] = np.uint64(st.rays); out["paths"] = np.uint64(st.paths)
        if rank == 0:
            out["frame"] = t.download_composed()
        out["calls"] = np.array([coll.calls[k] for k in ("reduce", "allreduce", "allgather", "alltoall", "gather")])
    elif case == "config4":
        # BASELINE config 4 as an N-rank tile split at the named size: Cornell + teapot.obj x 64 (1 005 056 triangles), tracePath
        W, H, spp = 1920, 1080, 2
        scene = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.golden("teapot").replicate(8, 80.0))
        t.upload_scene(scene.view); t.set_camera(host.prepare_camera(W, H)); t.set_environment((0.0, 0.0, 0.0)); t.resize(W, H)
        t.set_collectives(coll, world, rank)
        t.clear_accum(); t.seed(0x5EED0004); t.render(spp=spp, tile_rank=rank, tile_nranks=world); t.group_reduce_accum(0)
        st = t.stats()
        out["rays"] = np.uint64(st.rays); out["paths"] = np.uint64(st.paths)
        own = owner_mask(W, H, world, rank)
        rng = t.download_rng()
        out["rng_own_sha"] = np.array(sha(rng[own]))
        if rank == 0:
            out["frame"] = t.download_accum()
        # 8 spp more through the persistent-workgroup kernel (>= 8 spp on a tree read from memory), pipelined compose
        t.clear_accum(); t.seed(0x5EED0005); t.render(spp=8, tile_rank=rank, tile_nranks=world); t.group_reduce_accum_async(0)
        if rank == 0:
            out["frame8"] = t.download_composed()
    elif case == "config5":
        # BASELINE config 5 as an N-rank split at the named size: SPPM on the config-2 scene, 512^2 photons per frame
        W, H, frames = 1920, 1080, 4
        scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
        t.upload_scene(scene.view); t.set_camera(host.prepare_camera(W, H)); t.set_environment((0.0, 0.0, 0.0)); t.resize(W, H)
        t.set_collectives(coll, world, rank)
        t.clear_accum(); t.seed(0x5EED0050); t.sppm_init(0x5EED0051); t.sppm_frames(frames)
        cam, pho, mark, count, cx = t.sppm_download()
        t.group_reduce_accum(0)
        own = owner_mask(W, H, world, rank).ravel()
        out["cam_own_sha"] = np.array(sha(cam[own]))
        out["pho_sha"] = np.array(sha(pho)); out["mark_sha"] = np.array(sha(mark)); out["count_sha"] = np.array(sha(count))
        out["total"] = np.float32(cx.totalPhotonSum); out["hash_scale"] = np.float32(cx.photonHashScale)
        rng = t.download_rng().reshape(-1, 4)
        out["rng_own_sha"] = np.array(sha(rng[own]))
        if rank == 0:
            out["frame"] = t.download_accum()
        out["calls"] = np.array([coll.calls[k] for k in ("reduce", "allreduce", "allgather", "alltoall", "gather")])
    else:
        raise SystemExit(f"unknown case {case}")
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    t.group_finalize()
    dist.barrier()
    dist.destroy_process_group()
    t.close()


if __name__ == "__main__":
    main()
