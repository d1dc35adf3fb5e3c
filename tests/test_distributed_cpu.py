"""CPU, world_size 2, gloo: the N>1 path of the driver (broadcast of shared conditioning, dialogue sharding,
all_gather of final latents, max-over-ranks timing) — the same code bench.py runs over RCCL on GPUs."""
import os
import socket

import pytest

import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    from theatergen_amd import distributed as D
    from theatergen_amd import story
    r, w, _ = D.init(backend="gloo")
    assert (r, w) == (rank, world) and D.is_dist()
    shared = story.shared_conditioning(32, 4, torch.float32, "cpu")
    want = {k: v.clone() for k, v in shared.items()}
    if rank != 0:
        for v in shared.values():
            v.zero_()
    D.broadcast_conditioning(shared, src=0)
    ok_bcast = all(torch.equal(shared[k], want[k]) for k in want)
    mine = D.shard(list(range(8)), rank, world)
    local = torch.stack([torch.full((4, 8, 8), float(d)) for d in mine])      # "final latents" of my dialogues
    allv = D.gather_latents(local)
    ids = allv[:, 0, 0, 0].tolist()
    t = D.max_over_ranks(1.0 + rank, torch.device("cpu"))
    D.barrier()
    out.put((rank, ok_bcast, mine, ids, t))
    torch.distributed.destroy_process_group()


def test_broadcast_shard_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_bcast, mine, ids, t in res:
        assert ok_bcast
        assert mine == [d for d in range(8) if d % 2 == rank]
        assert ids == [0.0, 2.0, 4.0, 6.0, 1.0, 3.0, 5.0, 7.0]       # rank-major gather
        assert t == 2.0                                               # MAX over ranks


def test_bench_self_launches_its_ranks_world2():
    """`python bench.py --gpus 2` with no torchrun environment must START two ranks itself (it used to run one
    rank and print n_gpus = 1); the launcher is exercised on CPU with the gloo dry run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                  # rank 0 prints ONE line
    rec = json.loads(lines[0])
    assert rec == {"dry": True, "n_gpus": 2, "collectives_ok": True}
    # a torchrun environment that disagrees with --gpus is an error, not a silent 1-GPU run
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch"], env=env2,
                        capture_output=True, text=True, timeout=120)
    assert r2.returncode != 0 and "WORLD_SIZE=1" in (r2.stderr + r2.stdout)


@pytest.mark.parametrize("world", [4, 8])
def test_bench_dry_launch_world_4_and_8_through_the_real_launcher(world):
    """`python bench.py --gpus 8 --dry-launch` brings up 8 gloo ranks through the launcher the driver's SCALE run uses
    (self-launch under torch.distributed.run on 127.0.0.1), every rank pins its host threads to its own CPU slice, the timed region's
    collectives work and `unshard(all_gather(shard(items)))` is the identity at that world size."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--dry-launch"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["dry"] and rec["n_gpus"] == world and rec["collectives_ok"] and rec["unshard_of_shard_is_identity"], rec
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if rec["cpus_of_rank0"] is not None and ncpu >= world:
        assert rec["cpus_of_rank0"] == ncpu // world, rec            # rank 0 kept 1 / world of the CPUs the launcher could use


def test_unshard_inverts_shard_for_every_world_size():
    from theatergen_amd import distributed as D
    for world in (1, 2, 4, 8):
        items = list(range(8))
        gathered = torch.tensor([it for r in range(world) for it in D.shard(items, r, world)], dtype=torch.float32).reshape(-1, 1)
        assert [int(v) for v in D.unshard(gathered, world)[:, 0]] == items


def _strong_inputs_and_denoise(shared, img_tok, cidx, ctx):
    """CPU stand-ins with the engine's interface: conditioning by the bench's own story.job_conditioning, a per-image deterministic
    'denoiser' (every image computed on its own, so the bits cannot depend on which rank or batch an image lands in)."""
    from theatergen_amd import story

    def make_inputs(jobs):
        enc = story.job_conditioning(jobs, shared, img_tok, cidx, ctx, torch.float32, "cpu")
        lat = torch.stack([torch.randn((4, 8, 8), generator=torch.Generator().manual_seed(j.bg_seed * 7 + j.char)) for j in jobs])
        return enc, lat

    def denoise(enc, lat):
        n = lat.shape[0]
        out = []
        for i in range(n):
            neg, pos = enc[i], enc[n + i]
            x = lat[i]
            for _ in range(3):
                x = torch.tanh(x * pos[-4:].mean() + neg.std()) + 0.1 * pos[:77].mean(0)[:8].reshape(1, 1, 8)
            out.append(x)
        return torch.stack(out)

    return make_inputs, denoise


def _strong_worker(rank, world, port, out):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    from theatergen_amd import distributed as D
    from theatergen_amd import story
    D.init(backend="gloo")
    ctx, T = 32, 4
    jobs = story.story_jobs(2)
    shared = story.shared_conditioning(ctx, T, torch.float32, "cpu")
    char_ids = sorted({j.char_id for j in jobs})
    img_tok = story.character_image_tokens(char_ids, ctx, T, torch.float32, "cpu")
    if rank != 0:                       # only rank 0's copy counts: what the others hold before the broadcast must not matter
        for v in shared.values():
            v.fill_(123.0)
        img_tok.fill_(-7.0)
    D.broadcast_conditioning(shared, src=0)
    D.broadcast_conditioning({"image_tokens": img_tok}, src=0)
    cidx = {c: i for i, c in enumerate(char_ids)}
    make_inputs, denoise = _strong_inputs_and_denoise(shared, img_tok, cidx, ctx)
    got = D.run_story_strong(jobs, rank, world, make_inputs, denoise)
    mine = [(j.turn, j.char) for j in D.shard(jobs, rank, world)]
    out.put((rank, mine, got))
    torch.distributed.destroy_process_group()


def test_strong_scaling_partition_equals_single_rank_bit_for_bit():
    """SURVEY §8(e), second partitioning: one story's 8 (turn, character) jobs over 2 ranks, image tokens
    broadcast, final latents all-gathered and un-sharded == the single-rank result, bit for bit (reference independence argument:
    theatergen.py:214-271, one independent generation per (turn, character))."""
    from theatergen_amd import distributed as D
    from theatergen_amd import story
    ctx_, T = 32, 4
    jobs = story.story_jobs(2)
    shared = story.shared_conditioning(ctx_, T, torch.float32, "cpu")
    char_ids = sorted({j.char_id for j in jobs})
    img_tok = story.character_image_tokens(char_ids, ctx_, T, torch.float32, "cpu")
    cidx = {c: i for i, c in enumerate(char_ids)}
    make_inputs, denoise = _strong_inputs_and_denoise(shared, img_tok, cidx, ctx_)
    single = D.run_story_strong(jobs, 0, 1, make_inputs, denoise)            # no process group: the whole story on one rank
    assert single.shape == (8, 4, 8, 8)
    # unshard is the exact inverse of the round-robin shard
    ids = torch.arange(8.0)
    gathered = torch.cat([ids[r::2] for r in range(2)])
    assert torch.equal(D.unshard(gathered, 2), ids)
    gathered4 = torch.cat([ids[r::4] for r in range(4)])
    assert torch.equal(D.unshard(gathered4, 4), ids)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_strong_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [(1, 0), (2, 0), (3, 0), (4, 0)] and res[1][1] == [(1, 1), (2, 1), (3, 1), (4, 1)]      # rank = character here
    for rank, _, got in res:
        assert torch.equal(got, single), f"rank {rank}: gathered story differs from the single-rank result"
