"""Multi-GPU driver: one process per GPU, characters / dialogues sharded with NO data-path collective.

The reference has no distributed code at all (zero ``torch.distributed`` call sites); every
(dialogue, turn, character) stage-1 generation is an independent 50-step chain (reference ``theatergen.py:214-271``,
``generate.py:183-191``), so the path shards by dialogue with replicated weights.  RCCL over xGMI
(``torch.distributed`` backend "nccl" on ROCm) is used only for
  * ``broadcast`` of the shared conditioning (negative-prompt text embeds, per-character image tokens) from rank 0,
  * ``all_gather`` of the final latents (32-64 KB per image) at the end of a step.
Two partitionings: by DIALOGUE (weak scaling: every rank denoises its own story per step — the SCALE line) and
WITHIN a dialogue by (turn, character) (``run_story_strong``: one story's 8 jobs split over the ranks, `bench.py --scaling strong`).
Messages are KB-scale and latency-bound; there is no all-reduce and no ring.  The same code runs on CPU tensors
with the ``gloo`` backend (world_size-2 tests in tests/test_distributed_cpu.py).
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None, device=None):
    """Initialise the process group from the torchrun environment (RANK / WORLD_SIZE / MASTER_*)."""
    rank, world, local = env_world()
    # a torchrun environment (RANK set) gets its process group even at world size 1: the N = 1 leg of a scaling run then goes
    # through the same RCCL initialisation (library load, ``device_id`` binding) as N > 1; the collectives below are no-ops there
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if (world > 1 or launched) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def pin_host_threads(local, world):
    """One process per GPU on ONE node: N Python launch threads (+ torch's intra-op pools, used by the CPU-side latent recipe) must not fight over
    the same cores.  Rank ``local`` of ``world`` takes a contiguous slice of the CPUs this process may run on (``sched_setaffinity``) and sizes
    torch's pool to it.  Returns (first cpu, count).  A single-rank run keeps the whole machine.  The slice count is the number of ranks ON THIS NODE
    (``LOCAL_WORLD_SIZE`` of the launcher; the global ``world`` only when that is not set), so a multi-node launch still splits a node's CPUs among its
    own ranks.  The pinning is for the life of the process (every later CPU leg and subprocess inherits it): bench.py and the launcher tests are the
    only callers, one call per process."""
    try:
        world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    except ValueError:
        pass
    if world <= 1:
        return None
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cpus = list(range(os.cpu_count() or 1))
    per = max(1, len(cpus) // world)
    mine = cpus[(local % world) * per:(local % world) * per + per] or cpus
    try:
        os.sched_setaffinity(0, mine)
    except (AttributeError, OSError):
        pass
    torch.set_num_threads(max(1, min(len(mine), 8)))
    return mine[0], len(mine)


def is_dist(force=False):
    """True when collectives have to run: a process group of more than one rank — or, with ``force``, any initialised group
    (the world-size-1 RCCL test exercises the real library calls on one GPU)."""
    return dist.is_available() and dist.is_initialized() and (force or dist.get_world_size() > 1)


def shard(items, rank, world):
    """Round-robin partition of independent work items (dialogues / character jobs)."""
    return [it for i, it in enumerate(items) if i % world == rank]


def unshard(gathered, world):
    """Inverse of ``shard`` for an ``all_gather`` result: ``gathered`` [world * per, ...] is rank-major (rank r's items r, r + world,
    ...), the result is in ITEM order (item k * world + r = gathered[r * per + k]).  Needs the same count on every rank."""
    per = gathered.shape[0] // world
    assert per * world == gathered.shape[0]
    return gathered.reshape(world, per, *gathered.shape[1:]).transpose(0, 1).reshape(gathered.shape)


def run_story_strong(jobs, rank, world, make_inputs, denoise, force=False):
    """SECOND partitioning of SURVEY §8(e) — strong scaling WITHIN one dialogue: the story's (turn, character) jobs are independent
    50-step chains (reference theatergen.py:214-271: one `generate_single_object_with_box` call each, nothing shared but the
    character's image tokens), so rank r denoises jobs r, r + world, ...; the final latents are all-gathered and put back in job
    order.  ``make_inputs(my_jobs) -> (enc, lat)`` builds the conditioning from the (already broadcast) shared tensors;
    ``denoise(enc, lat) -> finals [n_local, C, h, w]``.  No data-path collective: one all_gather of 64 KB per image at the end."""
    if len(jobs) % world:
        raise ValueError(f"{len(jobs)} character jobs do not split evenly over {world} ranks")
    mine = shard(jobs, rank, world)
    enc, lat = make_inputs(mine)
    finals = denoise(enc, lat)
    if not is_dist(force):
        return finals
    return unshard(gather_latents(finals, force), dist.get_world_size())


def broadcast_conditioning(tensors, src=0, force=False):
    """In-place broadcast of a dict of tensors (shared text / image embeddings) from ``src``."""
    if is_dist(force):
        for k in sorted(tensors):
            dist.broadcast(tensors[k], src=src)
    return tensors


def gather_latents(local, force=False):
    """all_gather of per-rank final latents [n_local, C, h, w] -> [world * n_local, C, h, w] (rank-major)."""
    if not is_dist(force):
        return local
    world = dist.get_world_size()
    local = local.contiguous()
    out = torch.empty((world * local.shape[0], *local.shape[1:]), dtype=local.dtype, device=local.device)
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(out, local)
    else:
        parts = list(out.chunk(world, dim=0))
        dist.all_gather(parts, local)
    return out


def barrier(force=False):
    if is_dist(force):
        dist.barrier()


def max_over_ranks(value, device, force=False):
    if not is_dist(force):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
