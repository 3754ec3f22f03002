"""Row ownership and exchange layouts of the multi-GPU PEA forward (one process per GPU).

Destination rows are owned tile-interleaved: row i belongs to rank (i // tile) % world.  Every rank therefore owns
the same share of every node type (ids are contiguous per-type blocks, reference datasets/movielens.py:184-227) and
of both conv layers, so (a) the heavy relations shard evenly without a cost model, (b) the channel fusion
(reference models/base.py:196-203) is row-local.  Two kinds of exchange remain, both all-gathers:

  * gather sources of a level >= 1: the rows a relation reads (its distinct source nodes), produced by their owners,
    packed rank-major into an exchange buffer [world * M, ld] and all-gathered in place; the plan renames the CSR's
    source ids to slots of that buffer (pea_plan_set_sources), so kernels index it directly;
  * the fused [N, R] table before scoring (instead of the reference-literal [P, N, R] stack: same arithmetic,
    P times fewer bytes over xGMI).

The layout math is torch index arithmetic (device agnostic) and runs once, at plan time.  Per step, on CUDA tensors,
rows are packed into / unpacked from preallocated rank-major buffers by the HIP kernels of csrc/exchange.hip on the
launch stream and the all-gather runs IN PLACE on that buffer (no staging allocation, no clone); on CPU tensors the same
moves are torch index ops, so layouts and exchanges are exercised with the gloo backend (tests/test_sharding_cpu.py).
On the GPU box the collectives run over RCCL (backend 'nccl').
"""
import torch
import torch.distributed as dist


def _hip():
    from . import _lib
    return _lib


class CommTimer:
    """Optional device-time accounting of the exchanges (bench.py): pairs of CUDA events around every collective of
    this process, read back after a synchronize.  Off by default."""
    enabled = False
    events = []

    @classmethod
    def span(cls, device):
        if not cls.enabled or not torch.cuda.is_available() or torch.device(device).type != 'cuda':
            return None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cls.events.append((a, b))
        a.record()
        return b

    @classmethod
    def total_ms(cls):
        ms = sum(a.elapsed_time(b) for a, b in cls.events)
        cls.events = []
        return ms


class SourceLayout:
    """Exchange layout of one relation's source rows."""

    def __init__(self, shard, src_nodes):
        self.shard = shard
        self.src_nodes = src_nodes                                  # int64, ascending, global (same on every rank)
        owners = shard.owner(src_nodes)
        counts = torch.bincount(owners, minlength=shard.world)
        self.counts = [int(c) for c in counts.tolist()]
        self.slots_per_rank = (max(self.counts + [0]) + 7) // 8 * 8  # M, padded
        # index of each source among its owner's sources (src_nodes is ascending, so a stable sort by owner keeps
        # id order inside an owner)
        order = torch.argsort(owners, stable=True)
        starts = torch.cumsum(counts, 0) - counts
        local = torch.empty_like(order)
        local[order] = torch.arange(order.numel(), device=order.device) - starts[owners[order]]
        slots = owners * self.slots_per_rank + local
        self.slot_of_node = torch.full((shard.num_nodes,), -1, dtype=torch.int32, device=src_nodes.device)
        self.slot_of_node[src_nodes] = slots.to(torch.int32)
        mine = owners == shard.rank
        self.own_nodes = src_nodes[mine]                            # ascending == slot order inside this rank
        self.own_nodes_i32 = self.own_nodes.to(torch.int32).contiguous()   # what the HIP pack kernel reads
        self.own_count = int(self.own_nodes.numel())
        self.need_rows = torch.unique(torch.cat([shard.owned_rows(src_nodes.device), src_nodes])).to(torch.int32)
        # fill-in exchanges (sharded backward): the source rows OTHER ranks own, and their slots in the rank-major buffer
        self.other_nodes = src_nodes[~mine]
        self.other_nodes_i32 = self.other_nodes.to(torch.int32).contiguous()
        self.other_slots = slots[~mine]
        self.other_slots_i32 = self.other_slots.to(torch.int32).contiguous()
        self._bufs = {}

    def buffer(self, width, like):
        """Cached rank-major exchange buffer [world * M, width] for rows of `like`'s dtype / device."""
        key = (int(width), like.dtype, str(like.device))
        if key not in self._bufs:
            self._bufs[key] = torch.zeros((self.shard.world * self.slots_per_rank, int(width)), dtype=like.dtype, device=like.device)
        return self._bufs[key]


class ShardLayout:
    _inplace_checked = False
    _inplace_ok = True   # set by verify_inplace_all_gather (start-up comparison with the out-of-place form)

    def __init__(self, num_nodes, rank=0, world=1, tile=256):
        if not (0 <= rank < world) or tile <= 0:
            raise ValueError('bad shard (rank %d of %d, tile %d)' % (rank, world, tile))
        self.num_nodes, self.rank, self.world, self.tile = int(num_nodes), int(rank), int(world), int(tile)
        self._owned = {}
        self._gather_plan = {}
        self._flags = {}
        self.dry = False   # True: skip the collectives (single-process rehearsal of one rank's compute + host work)
        self.dry_log = []  # ... and note what each would have moved: (kind, bytes this rank would receive / reduce)

    @classmethod
    def verify_inplace_all_gather(cls, device, group=None, rows=4096):
        """Start-up self-check (backend 'nccl' = RCCL, world > 1): the in-place all-gather form every exchange uses
        (send block = this rank's slice of the receive buffer) is run once on a small buffer and COMPARED with the
        out-of-place form; `_inplace_ok` is set from that comparison on all ranks together (MIN over the ranks), not from an
        exception -- an aliasing problem that shows up as wrong data, or asynchronously, would otherwise go unnoticed.
        Returns the verdict.  Other backends (gloo rehearsal) do not alias buffers: nothing to check."""
        if not dist.is_available() or not dist.is_initialized():
            return cls._inplace_ok
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        if world == 1 or dist.get_backend(group) != 'nccl' or cls._inplace_checked:
            return cls._inplace_ok
        cls._inplace_checked = True
        mine = (torch.arange(rows, device=device, dtype=torch.float32) + float(rank * rows)) * 0.5
        want = torch.empty(world * rows, dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(want, mine, group=group)
        buf = torch.full((world * rows,), -1.0, dtype=torch.float32, device=device)
        buf[rank * rows:(rank + 1) * rows] = mine
        ok = True
        try:
            dist.all_gather_into_tensor(buf, buf[rank * rows:(rank + 1) * rows], group=group)
            torch.cuda.synchronize(device)
            ok = bool(torch.equal(buf, want))
        except RuntimeError:
            ok = False
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        cls._inplace_ok = bool(int(flag.item()))
        return cls._inplace_ok

    def owner(self, nodes):
        return torch.div(nodes, self.tile, rounding_mode='floor') % self.world

    def rows_of(self, rank, device='cpu'):
        rows = torch.arange(self.num_nodes, device=device)
        return rows[self.owner(rows) == rank]

    def owned_rows(self, device='cpu'):
        key = str(device)
        if key not in self._owned:
            self._owned[key] = self.rows_of(self.rank, device)
        return self._owned[key]

    def source_layout(self, edge_index):
        """edge_index: int64 [2, E] (row 0 = source).  Global, identical on every rank."""
        return SourceLayout(self, torch.unique(edge_index[0]))

    # ---------------------------------------------------------------- collectives
    def _all_gather_blocks(self, buf, block_rows, group=None, async_op=False):
        """buf: contiguous [world * block_rows, ld]; block `rank` holds this rank's rows, the others are filled in.
        async_op (RCCL only): the collective is issued on RCCL's own stream behind the work already enqueued on the
        current stream and a work handle is returned; kernels launched afterwards run beside it until wait_all()."""
        if self.dry and self.world > 1 and block_rows:
            self.dry_log.append(('all_gather', (self.world - 1) * block_rows * buf.stride(0) * buf.element_size()))
        if self.world == 1 or block_rows == 0 or self.dry:
            return None
        mine = buf[self.rank * block_rows:(self.rank + 1) * block_rows]
        if async_op and dist.get_backend(group) == 'nccl':
            send = mine.reshape(-1) if ShardLayout._inplace_ok else mine.reshape(-1).clone()
            return dist.all_gather_into_tensor(buf.view(-1), send, group=group, async_op=True)
        done = CommTimer.span(buf.device)
        if dist.get_backend(group) == 'nccl':
            # RCCL all-gather IN PLACE: the send block is this rank's slice of the receive buffer (ncclAllGather's
            # documented in-place form, sendbuff == recvbuff + rank * count).  Should a torch / RCCL build refuse aliased
            # buffers, the send block is copied out once and for all later calls (slower by one copy, same result).
            # `_inplace_ok` comes from verify_inplace_all_gather (a data comparison at start-up, all ranks agree).
            if ShardLayout._inplace_ok:
                dist.all_gather_into_tensor(buf.view(-1), mine.reshape(-1), group=group)
            else:
                dist.all_gather_into_tensor(buf.view(-1), mine.reshape(-1).clone(), group=group)
        else:                                                                          # gloo (CPU tests / rehearsal)
            parts = [torch.empty(mine.shape, dtype=buf.dtype) for _ in range(self.world)]
            dist.all_gather(parts, mine.detach().cpu().contiguous(), group=group)
            for r, p in enumerate(parts):
                if r != self.rank:
                    buf[r * block_rows:(r + 1) * block_rows].copy_(p)
        if done is not None:
            done.record()
        return None

    def wait_all(self, works):
        """The current stream waits for the collectives started with async_op (device-side wait: the host does not
        block).  The time the stream actually spends waiting -- the part of the exchange that the work issued in between
        did not cover -- is what CommTimer records for them."""
        works = [w for w in works if w is not None]
        if not works:
            return
        done = CommTimer.span(torch.cuda.current_device()) if torch.cuda.is_available() else None
        for w in works:
            w.wait()
        if done is not None:
            done.record()

    def exchange_sources(self, xbuf, table, layout, col, width, group=None, packed=False, async_op=False):
        """xbuf [world*M, ld] <- all-gather of table[owner's source nodes, col:col+width] (slot order).  packed: this rank's
        block of xbuf is already filled (the producing kernel wrote the rows straight into their slots)."""
        m = layout.slots_per_rank
        if m == 0:
            return None
        if layout.own_count and not packed:
            if table.is_cuda:
                lib = _hip()
                lib.check(lib.load().pea_rows_pack(lib.ptr(table), table.stride(0), int(col), int(width),
                                                   lib.ptr(layout.own_nodes_i32), layout.own_count,
                                                   lib.ptr(xbuf[self.rank * m:]), xbuf.stride(0), lib.current_stream()))
            else:
                xbuf[self.rank * m:self.rank * m + layout.own_count, :width] = table[layout.own_nodes, col:col + width]
        return self._all_gather_blocks(xbuf, m, group, async_op=async_op)

    def reduce_rows(self, rows, group=None):
        """Sum over the ranks of rows [K, W] in which every rank filled the entries it owns and zeroed the others (x + 0 is
        exact): the second half of gather_rows for rows a kernel already selected."""
        if self.dry and self.world > 1:
            self.dry_log.append(('all_reduce', rows.numel() * rows.element_size()))
        if self.world == 1 or self.dry:
            return rows
        done = CommTimer.span(rows.device)
        if dist.get_backend(group) == 'nccl':
            dist.all_reduce(rows, group=group)
        else:
            host = rows.detach().cpu()
            dist.all_reduce(host, group=group)
            rows = host.to(rows.device)
        if done is not None:
            done.record()
        return rows

    def reduce_rows_(self, rows, group=None):
        """reduce_rows in place (the buffer a captured graph reads next must keep its address)."""
        out = self.reduce_rows(rows, group)
        if out is not rows:
            rows.copy_(out)
        return rows

    def gather_rows(self, table, ids, group=None):
        """[len(ids), ...] rows table[ids] where every rank only holds the rows it owns: each rank contributes its own
        rows (zeros elsewhere) and the contributions are summed -- one small all-reduce instead of the all-gather of
        the whole table when only a batch of rows is needed (the BPR triples of a training step).  x + 0 is exact."""
        if self.world == 1:
            return table[ids]
        if table.is_cuda and table.dim() == 2 and table.shape[1] % 4 == 0 and table.stride(1) == 1 and ids.dim() == 1:
            # one HIP launch: rows this rank owns, zeros elsewhere (an id out of range sets the flag; IndexError at the
            # next engine.check_pending_errors(), like a bad BPR triple)
            lib = _hip()
            ids = ids if ids.dtype == torch.int64 else ids.to(torch.int64)
            rows = torch.empty((ids.numel(), table.shape[1]), dtype=table.dtype, device=table.device)
            flag = self._err_flag(table.device)
            lib.check(lib.load().pea_rows_select_owned(lib.ptr(table), table.stride(0), table.shape[1], table.shape[0],
                                                       lib.ptr(ids), ids.stride(0), ids.numel(), self.rank, self.world,
                                                       self.tile, lib.ptr(rows), lib.ptr(flag), lib.current_stream()))
        else:
            rows = table[ids]
            mine = self.owner(ids) == self.rank   # rows of other ranks are undefined here (may hold NaN): select, never scale
            rows = torch.where(mine.view(-1, *([1] * (rows.dim() - 1))), rows, torch.zeros((), dtype=rows.dtype, device=rows.device))
        if self.dry:
            self.dry_log.append(('all_reduce', rows.numel() * rows.element_size()))
            return rows
        done = CommTimer.span(rows.device)
        if dist.get_backend(group) == 'nccl':
            dist.all_reduce(rows, group=group)
        else:
            host = rows.detach().cpu()
            dist.all_reduce(host, group=group)
            rows = host.to(rows.device)
        if done is not None:
            done.record()
        return rows

    def fill_in_rows(self, table, col, width, layout, group=None):
        """table [N, ld]: columns [col, col + width) of the rows at layout.src_nodes -- every rank has written the ones it
        owns; fills in the others' IN PLACE (pack own rows -> in-place all-gather -> unpack the other ranks' rows).  The
        sharded backward uses it on the node-indexed gradient buffers: the gathers over a reversed relation read rows of
        the forward relation's destinations, which their owners produced."""
        m = layout.slots_per_rank
        if self.world == 1 or m == 0:
            return
        buf = layout.buffer(width, table)
        hip = table.is_cuda and table.dtype == torch.float32 and width % 4 == 0 and col % 4 == 0 and table.stride(1) == 1 \
            and table.stride(0) % 4 == 0
        lib = _hip() if hip else None
        if layout.own_count:
            if hip:
                lib.check(lib.load().pea_rows_pack(lib.ptr(table), table.stride(0), int(col), int(width),
                                                   lib.ptr(layout.own_nodes_i32), layout.own_count,
                                                   lib.ptr(buf[self.rank * m:]), buf.stride(0), lib.current_stream()))
            else:
                buf[self.rank * m:self.rank * m + layout.own_count] = table[layout.own_nodes, col:col + width]
        self._all_gather_blocks(buf, m, group)
        if layout.other_nodes.numel():
            if hip:
                lib.check(lib.load().pea_rows_unpack(lib.ptr(buf), buf.stride(0), lib.ptr(layout.other_slots_i32), int(width),
                                                     lib.ptr(layout.other_nodes_i32), layout.other_nodes_i32.numel(),
                                                     lib.ptr(table), table.stride(0), int(col), lib.current_stream()))
            else:
                table[layout.other_nodes, col:col + width] = buf[layout.other_slots]

    def fill_in_rows_batch(self, items, group=None):
        """fill_in_rows for a list of (table, col, width, layout) with ONE pack launch, ONE all-gather and ONE unpack launch
        (pea_rows_pack_batch / _unpack_batch on a staging buffer whose rank block holds every item's rows one after the
        other): a sharded backward level fills in all its gradient buffers at once instead of one exchange per relation
        and buffer.  CPU tensors (gloo rehearsal) take the per-item path."""
        items = [it for it in items if it[3].slots_per_rank > 0]
        if self.world == 1 or not items:
            return
        hip = all(t.is_cuda and t.dtype == torch.float32 and w % 4 == 0 and c % 4 == 0 and t.stride(1) == 1 and t.stride(0) % 4 == 0
                  for t, c, w, _ in items)
        if not hip or len(items) == 1:
            for t, c, w, lay in items:
                self.fill_in_rows(t, c, w, lay, group)
            return
        lib = _hip()
        block = sum(lay.slots_per_rank * w for _, _, w, lay in items)          # floats per rank
        dev = items[0][0].device
        key = (str(dev), self.world * block)
        stage = self._stage.get(key) if hasattr(self, '_stage') else None
        if stage is None:
            if not hasattr(self, '_stage'):
                self._stage = {}
            stage = self._stage[key] = torch.empty((self.world, block), dtype=torch.float32, device=dev)
        n = len(items)
        pack = (lib.XchgJob * n)()
        unpack = (lib.XchgJob * n)()
        off = 0
        for q, (t, c, w, lay) in enumerate(items):
            pack[q] = lib.XchgJob(t.data_ptr(), t.stride(0), int(c), int(w), lay.own_nodes_i32.data_ptr() if lay.own_count else None,
                                  None, int(lay.own_count), off, int(lay.slots_per_rank))
            k = lay.other_nodes_i32.numel()
            unpack[q] = lib.XchgJob(t.data_ptr(), t.stride(0), int(c), int(w), lay.other_nodes_i32.data_ptr() if k else None,
                                    lay.other_slots_i32.data_ptr() if k else None, int(k), off, int(lay.slots_per_rank))
            off += lay.slots_per_rank * w
        lib.check(lib.load().pea_rows_pack_batch(n, pack, lib.ptr(stage[self.rank]), lib.current_stream()))
        self._all_gather_blocks(stage, 1, group)
        lib.check(lib.load().pea_rows_unpack_batch(n, unpack, lib.ptr(stage), block, lib.current_stream()))

    def fill_in_ids(self, table, col, width, ids, group=None):
        """The same for an explicit list of node ids (duplicates allowed; e.g. the rows of a BPR batch): every rank
        contributes the listed rows it owns, one all-reduce of [len(ids), width] (x + 0: exact), and every listed row of
        `table` is overwritten with the owner's values."""
        if self.world == 1 or ids.numel() == 0:
            return
        rows = self.gather_rows(table[:, col:col + width], ids, group)
        if table.is_cuda and table.dtype == torch.float32 and width % 4 == 0 and col % 4 == 0 and table.stride(1) == 1:
            lib = _hip()
            ids32 = ids.to(torch.int32)
            lib.check(lib.load().pea_rows_unpack(lib.ptr(rows), rows.stride(0), None, int(width), lib.ptr(ids32), ids32.numel(),
                                                 lib.ptr(table), table.stride(0), int(col), lib.current_stream()))
        else:
            table[ids, col:col + width] = rows

    def all_reduce_sum_(self, tensors, group=None):
        """Sum every tensor of the list over the ranks, in place, with ONE collective on a flat copy (the ranks' shares of
        the parameter gradients of a sharded training step)."""
        tensors = [t for t in tensors if t is not None and t.numel()]
        if self.dry and self.world > 1 and tensors:
            self.dry_log.append(('all_reduce', sum(t.numel() * t.element_size() for t in tensors)))
        if self.world == 1 or not tensors or self.dry:
            return
        flat = torch.cat([t.reshape(-1) for t in tensors])
        done = CommTimer.span(flat.device)
        if dist.get_backend(group) == 'nccl':
            dist.all_reduce(flat, group=group)
        else:
            host = flat.detach().cpu()
            dist.all_reduce(host, group=group)
            flat = host.to(flat.device)
        if done is not None:
            done.record()
        off = 0
        for t in tensors:
            t.copy_(flat[off:off + t.numel()].view(t.shape))
            off += t.numel()

    def _err_flag(self, device):
        from . import engine
        key = str(device)
        if key not in self._flags:
            self._flags[key] = torch.zeros(1, dtype=torch.int32, device=device)
        flag = self._flags[key]
        if not any(f is flag for f in engine._pending_err):
            engine._pending_err.append(flag)
        return flag

    def allgather_rows(self, table, group=None):
        """table [N, ...]: every rank has written the rows it owns; fills in everybody else's (in place).  The rank-major
        staging buffer and the row lists are built once per (device, row width) and reused."""
        if self.world == 1:
            return table
        dev = table.device
        flat = table.reshape(self.num_nodes, -1)
        width = flat.shape[1]
        key = (str(dev), width, table.dtype)
        if key not in self._gather_plan:
            rows = [self.rows_of(r, dev) for r in range(self.world)]
            m = max(int(x.numel()) for x in rows)
            dest = torch.cat([x for r, x in enumerate(rows) if r != self.rank])
            src = torch.cat([r * m + torch.arange(x.numel(), device=dev) for r, x in enumerate(rows) if r != self.rank])
            buf = torch.zeros((self.world * m, width), dtype=table.dtype, device=dev)
            self._gather_plan[key] = (m, dest, src, dest.to(torch.int32), src.to(torch.int32), buf,
                                      self.owned_rows(dev).to(torch.int32).contiguous())
        m, dest, src, dest32, src32, buf, own32 = self._gather_plan[key]
        own = self.owned_rows(dev)
        hip = flat.is_cuda and flat.dtype == torch.float32 and width % 4 == 0 and flat.stride(1) == 1 and flat.stride(0) % 4 == 0
        if hip:
            lib = _hip()
            lib.check(lib.load().pea_rows_pack(lib.ptr(flat), flat.stride(0), 0, width, lib.ptr(own32), own32.numel(),
                                               lib.ptr(buf[self.rank * m:]), buf.stride(0), lib.current_stream()))
        else:
            buf[self.rank * m:self.rank * m + own.numel()] = flat[own]
        self._all_gather_blocks(buf, m, group)
        if hip:
            lib.check(lib.load().pea_rows_unpack(lib.ptr(buf), buf.stride(0), lib.ptr(src32), width, lib.ptr(dest32),
                                                 dest32.numel(), lib.ptr(flat), flat.stride(0), 0, lib.current_stream()))
        else:
            flat[dest] = buf[src]
        return table
