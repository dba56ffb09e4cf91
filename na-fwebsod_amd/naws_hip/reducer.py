"""Gradient all-reduce schedule over the parameter arena (reference:
detectron/modeling/optimizer_wsl.py:52-72 issues one NCCLAllreduce per parameter blob).

One process per GPU; the backend is whatever the process group was created with ("nccl" =
RCCL over xGMI on MI355X, "gloo" in the CPU tests).  Two kinds of message:
  * one slice for every gradient except fc6_w (ready early in backward), and
  * fc6_w's gradient in row chunks, each handed over while the next chunk's wgrad GEMM runs,
so the 822 MB that backward produces last never sits exposed behind the compute.
"""


def row_chunks(rows, n_chunks, align=128):
    """[(r0, r1)) covering [0, rows) in at most n_chunks pieces whose size is a multiple of
    `align` (the GEMM's M-tile), so every piece is a full-tile wgrad launch."""
    n_chunks = max(1, int(n_chunks))
    step = (rows + n_chunks - 1) // n_chunks
    step = (step + align - 1) // align * align
    out, r0 = [], 0
    while r0 < rows:
        out.append((r0, min(rows, r0 + step)))
        r0 += step
    return out


def message_plan(arena, rows6, allreduce_chunks, active, pipelined=False):
    """The gradient messages of one iteration, in the order backward produces and hands them to
    the collective: fc6_w's gradient (86 % of the bytes) in row chunks - each chunk is reduced as
    soon as its wgrad GEMM is queued - then ONE message with every other gradient (fc6 / fc7
    biases, fc7_w, fc8*: contiguous in the arena).  -> [('fc6_w', (r0, r1)), ..., ('small', None)].
    pipelined (the N > 1 step whose fc6 forward starts piece by piece, engine._apply_update_pipelined):
    fc6's two bias vectors (32 KB, known before the weight gradient: a column sum of dZ6) travel
    FIRST as a message of their own, because the first forward piece of the next iteration needs
    them, and the last message holds the rest: [('fc6_b', None), ('fc6_w', (r0, r1)), ...,
    ('rest', None)].  Same bytes, same sums.
    The engine walks this list (engine._head_backward); tests/test_distributed_cpu.py replays it
    on a CPU arena over gloo."""
    plan = [('fc6_w', rc) for rc in row_chunks(rows6, allreduce_chunks if active else 1)]
    if pipelined:
        return [('fc6_b', None)] + plan + [('rest', None)]
    plan.append(('small', None))
    return plan


def message_slice(arena, grads, kind, rows, k6):
    """The contiguous arena slice a message of `message_plan` covers."""
    if kind == 'fc6_w':
        o6 = arena.offsets['fc6_w'][0]
        return grads[o6 + rows[0] * k6:o6 + rows[1] * k6]
    if kind == 'fc6_b':
        return arena.span(grads, 'fc6_b', '_[noisy]_fc6_b')
    if kind == 'rest':
        return arena.span(grads, 'fc7_w', 'noisy_fc8d_b')
    return arena.span(grads, 'fc6_b', 'noisy_fc8d_b')


def owner_blocks(rows, world, align=32):
    """fc6_w's rows cut into one contiguous block per rank (NAWS.SHARDED_UPDATE): rank r owns
    rows [r * rows / world, (r + 1) * rows / world) - its fp32 master rows, their momentum and
    their update.  None when the rows do not divide into `align`-multiples."""
    world = int(world)
    if world < 1 or rows % world != 0 or (rows // world) % align != 0:
        return None
    b = rows // world
    return [(r * b, (r + 1) * b) for r in range(world)]


def owner_pieces(r0, r1, blocks):
    """[(owner, p0, p1)]: the row range [r0, r1) of one gradient message cut at the owners'
    block boundaries."""
    out = []
    for o, (b0, b1) in enumerate(blocks):
        p0, p1 = max(r0, b0), min(r1, b1)
        if p0 < p1:
            out.append((o, p0, p1))
    return out


def state_digest(tensors):
    """int64 [2 * len(tensors)]: per buffer (sum of its words, position-weighted sum of its
    words), both exact integer sums of the raw bit patterns - equal on two ranks iff (up to a 2^-64
    accident) the buffers are bit-identical.  Chunked: the 957.7 MB arenas are never widened to
    int64 as a whole.  (Checkpoint-time / bench-time bookkeeping in plain torch ops: not on the
    step's path.)"""
    import torch
    out = []
    weights = {}
    for t in tensors:
        v = t.detach().contiguous().view(-1)
        v = v.view({1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[v.element_size()])
        s1 = torch.zeros((), dtype=torch.int64, device=v.device)
        s2 = torch.zeros((), dtype=torch.int64, device=v.device)
        step = 1 << 24
        for o in range(0, v.numel(), step):
            c = v[o:o + step].to(torch.int64)
            key = (c.numel(), str(v.device))
            if key not in weights:
                weights[key] = torch.arange(c.numel(), device=v.device, dtype=torch.int64) % 8191 + 1
            s1 += c.sum()
            s2 += (c * weights[key]).sum() + (o // step) * 7 * c.sum()
        out += [s1, s2]
    return torch.stack(out)


def ranks_agree(state, process_group, rank, world_size):
    """COLLECTIVE.  state: {name: tensor} (engine.state_tensors()) -> (True when every rank holds
    bit-identical buffers, [names of the buffers that differ]).  One all-reduce of a
    [world, 2 * buffers] int64 table.  The reference's ranks are GPUs of one process that start
    from broadcast blobs and apply the same all-reduced gradients (optimizer_wsl.py:52-72): equal
    by construction; here that property is CHECKED - after warm-up in bench.py, at every snapshot
    in the training loop."""
    import torch
    import torch.distributed as dist
    names, tensors = zip(*sorted(state.items()))
    dig = state_digest(tensors)
    table = torch.zeros((int(world_size), dig.numel()), dtype=torch.int64, device=dig.device)
    table[int(rank)] = dig
    dist.all_reduce(table, group=process_group)
    same = (table == table[0:1]).all(dim=0).cpu().tolist()
    bad = sorted({names[i // 2] for i, ok in enumerate(same) if not ok})
    return not bad, bad


class ArenaReducer(object):
    def __init__(self, process_group=None, world_size=1):
        self.pg = process_group
        self.world_size = int(world_size)
        self._pending = []
        self.force = False     # run the collectives even with a single rank (tests)
        # when a list: every message handed over is appended as [kind, elements] - the schedule a
        # run really issued (bench.py's exchange_messages_per_step, the world-8 test)
        self.log = None

    def _note(self, kind, t):
        if self.log is not None:
            self.log.append([kind, int(t.numel())])

    @property
    def active(self):
        return self.pg is not None and (self.world_size > 1 or self.force)

    def reduce_async(self, flat_slice):
        """Sum `flat_slice` (a contiguous view of the gradient arena) over all ranks, in place.
        Returns immediately; the collective is ordered after the work already queued on the
        caller's current stream."""
        if not self.active:
            return
        import torch.distributed as dist
        assert flat_slice.is_contiguous()
        self._note('all_reduce', flat_slice)
        self._pending.append(dist.all_reduce(flat_slice, op=dist.ReduceOp.SUM, group=self.pg,
                                             async_op=True))

    def wait(self):
        for w in self._pending:
            w.wait()
        self._pending = []

    def wait_first(self, n):
        """Wait for the `n` oldest messages still in flight only (hand-over order): the pipelined
        update consumes fc6_w's chunks one by one while the later ones are still on the links."""
        for w in self._pending[:n]:
            w.wait()
        del self._pending[:n]

    def in_flight(self):
        return len(self._pending)

    # ---- NAWS.SHARDED_UPDATE: gradient rows to their owner, updated rows back to everybody ----
    def _emulated(self, t):
        """gloo carries device tensors only through all_reduce and broadcast: the two-ranks-on-
        one-GPU test (and nothing else) takes the emulation below."""
        import torch.distributed as dist
        return t.is_cuda and dist.get_backend(self.pg) == 'gloo'

    def reduce_to_owner_async(self, flat_slice, owner):
        """Sum `flat_slice` over all ranks INTO rank `owner`'s copy (the other ranks' copies are
        left unspecified): one leg of the reduce-scatter of fc6_w's gradient rows."""
        if not self.active:
            return
        import torch.distributed as dist
        assert flat_slice.is_contiguous()
        self._note('reduce_to_owner', flat_slice)
        if self._emulated(flat_slice):
            w = dist.all_reduce(flat_slice, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        else:
            w = dist.reduce(flat_slice, dst=dist.get_global_rank(self.pg, owner),
                            op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self._pending.append(w)

    def gather_blocks_async(self, flat, rank):
        """`flat` = world equal contiguous blocks, block r valid on rank r -> every block valid
        everywhere, in place (all-gather; on gloo: one broadcast per owner)."""
        if not self.active:
            return
        import torch.distributed as dist
        assert flat.is_contiguous() and flat.numel() % self.world_size == 0
        self._note('gather_blocks', flat)
        n = flat.numel() // self.world_size
        if dist.get_backend(self.pg) == 'gloo':
            for o in range(self.world_size):
                self._pending.append(dist.broadcast(flat[o * n:(o + 1) * n],
                                                    src=dist.get_global_rank(self.pg, o),
                                                    group=self.pg, async_op=True))
        else:
            self._pending.append(dist.all_gather_into_tensor(flat, flat[rank * n:(rank + 1) * n],
                                                             group=self.pg, async_op=True))


class EmulatedExchange(object):
    """MEASUREMENT AID - `bench.py --emulate-exchange N[,CUS[,GB/s]]` on a one-GPU box: takes the
    reducer's place and, for every gradient message the engine hands over, queues what an
    N-rank ring all-reduce of that message does to THIS rank: a reduce-scatter phase and an
    all-gather phase, each moving (N-1)/N of the message through HBM on `cus` compute units at
    the pace the xGMI links allow, on a communication stream ordered behind the kernels that
    produced the message.  The gradients are only read (the copy lands in a scratch buffer), so
    the arithmetic is the one-rank step's; the schedule - chunked fc6_w messages during the
    backward pass, the deferred update waiting for the last of them underneath the next conv
    body - is the N-rank one.  The step time measured this way is a PROJECTION of the N-rank
    step (it has the contention for compute units and HBM, not the links' own hiccups)."""

    def __init__(self, device, n_ranks, cus=32, gbytes_per_sec=None):
        import torch
        self.device = device
        self.n = int(n_ranks)
        self.cus = int(cus)
        # default pace: min(N-1, 7) xGMI links x 153 GB/s per direction at 60 % efficiency
        self.gbps = float(gbytes_per_sec) if gbytes_per_sec else 0.6 * 153.0 * min(self.n - 1, 7)
        self.world_size = self.n       # (only NAWS.SHARDED_UPDATE's ownership reads it; the
        self.force = True              #  update's 1/gpu_num is the caller's business)
        # ONE communication stream per device for every instance: a process that builds several
        # exchanges one after another (bench.py's projections) would otherwise keep drawing new
        # streams, and HIP maps streams onto a handful of hardware queues round-robin
        # (GPU_MAX_HW_QUEUES) - two streams on one queue run in order, which serialises exactly
        # the overlap this class exists to measure
        key = str(device)
        if key not in EmulatedExchange._streams:
            # (default priority: a high-priority proxy stream projects worse at every rank count -
            # 18.8 / 15.9 / 15.7 vs 18.4 / 14.9 / 14.4 ms - it takes its CUs from the GEMMs sooner)
            EmulatedExchange._streams[key] = torch.cuda.Stream(device=device)
        self._stream = EmulatedExchange._streams[key]
        self._scratch = None
        self._events = []              # one per message in flight, recorded behind its proxy kernels
        self.total_bytes = 0           # moved since construction (bench: / steps)
        self.log = None                # as ArenaReducer.log

    active = True
    _streams = {}

    def _note(self, kind, t):
        if self.log is not None:
            self.log.append([kind, int(t.numel())])

    def reduce_async(self, flat_slice):
        self._note('all_reduce', flat_slice)
        self._phase(flat_slice, 2)         # reduce-scatter, all-gather

    def _scratch_for(self, nbytes):
        """The landing buffer of the proxy copies, at least `nbytes` long.  It is only ever
        touched on the communication stream, so it is allocated THERE (the caching allocator
        ties a block to the stream it was allocated on): a block that is outgrown goes back to
        that stream's pool and cannot be handed to a main-stream tensor while proxy kernels
        queued earlier still write into it."""
        import torch
        n = (nbytes + 3) // 4
        if self._scratch is None or self._scratch.numel() < n:
            with torch.cuda.stream(self._stream):
                self._scratch = torch.empty((n,), device=self.device, dtype=torch.float32)
        return self._scratch

    def _phase(self, flat, phases):
        import torch
        from . import ops
        nbytes = flat.numel() * flat.element_size()
        scratch = self._scratch_for(nbytes)
        part = int(nbytes * (self.n - 1) / self.n) // 16 * 16
        self._stream.wait_event(torch.cuda.current_stream(self.device).record_event())
        with torch.cuda.stream(self._stream):
            for _phase in range(phases):
                ops.emulate_exchange(flat, scratch, part, self.cus, self.gbps)
            self._events.append(self._stream.record_event())
        self.total_bytes += phases * part

    # NAWS.SHARDED_UPDATE under projection: this process plays rank 0 of N - it updates rows
    # [0, 8192 / N) of fc6_w only (the other rows simply stay as they are: the arithmetic of the
    # projection run is not a training run's), the reduce-scatter and the all-gather each move
    # (N-1)/N of their message
    def reduce_to_owner_async(self, flat_slice, owner):
        self._note('reduce_to_owner', flat_slice)
        self._phase(flat_slice, 1)

    def gather_blocks_async(self, flat, rank):
        self._note('gather_blocks', flat)
        if flat.dtype.is_floating_point and flat.numel() >= 4096:
            self._phase(flat, 1)

    def wait(self):
        import torch
        torch.cuda.current_stream(self.device).wait_event(self._stream.record_event())
        self._events = []

    def wait_first(self, n):
        import torch
        if n > 0 and self._events:
            n = min(n, len(self._events))
            torch.cuda.current_stream(self.device).wait_event(self._events[n - 1])
            del self._events[:n]

    def in_flight(self):
        return len(self._events)
