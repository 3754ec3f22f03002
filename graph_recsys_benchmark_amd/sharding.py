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

Everything here is torch index arithmetic (device agnostic) so the layouts and exchanges are exercised on CPU with
the gloo backend (tests/test_sharding_cpu.py); on the GPU box the same code runs over RCCL (backend 'nccl').
"""
import torch
import torch.distributed as dist


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
        self.own_count = int(self.own_nodes.numel())
        self.need_rows = torch.unique(torch.cat([shard.owned_rows(src_nodes.device), src_nodes])).to(torch.int32)


class ShardLayout:
    def __init__(self, num_nodes, rank=0, world=1, tile=256):
        if not (0 <= rank < world) or tile <= 0:
            raise ValueError('bad shard (rank %d of %d, tile %d)' % (rank, world, tile))
        self.num_nodes, self.rank, self.world, self.tile = int(num_nodes), int(rank), int(world), int(tile)
        self._owned = {}
        self._gather_plan = {}
        self.dry = False   # True: skip the collectives (single-process rehearsal of one rank's compute + host work)

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
    def _all_gather_blocks(self, buf, block_rows, group=None):
        """buf: contiguous [world * block_rows, ld]; block `rank` holds this rank's rows, the others are filled in."""
        if self.world == 1 or block_rows == 0 or self.dry:
            return
        mine = buf[self.rank * block_rows:(self.rank + 1) * block_rows]
        done = CommTimer.span(buf.device)
        if dist.get_backend(group) == 'nccl':
            # RCCL all-gather; the send block is copied out first so input and output never alias
            dist.all_gather_into_tensor(buf.view(-1), mine.reshape(-1).clone(), group=group)
        else:                                                                          # gloo (CPU tests / rehearsal)
            parts = [torch.empty(mine.shape, dtype=buf.dtype) for _ in range(self.world)]
            dist.all_gather(parts, mine.detach().cpu().contiguous(), group=group)
            for r, p in enumerate(parts):
                if r != self.rank:
                    buf[r * block_rows:(r + 1) * block_rows].copy_(p)
        if done is not None:
            done.record()

    def exchange_sources(self, xbuf, table, layout, col, width, group=None):
        """xbuf [world*M, ld] <- all-gather of table[owner's source nodes, col:col+width] (slot order)."""
        m = layout.slots_per_rank
        if m == 0:
            return
        if layout.own_count:
            xbuf[self.rank * m:self.rank * m + layout.own_count, :width] = table[layout.own_nodes, col:col + width]
        self._all_gather_blocks(xbuf, m, group)

    def gather_rows(self, table, ids, group=None):
        """[len(ids), ...] rows table[ids] where every rank only holds the rows it owns: each rank contributes its own
        rows (zeros elsewhere) and the contributions are summed -- one small all-reduce instead of the all-gather of
        the whole table when only a batch of rows is needed (the BPR triples of a training step).  x + 0 is exact."""
        rows = table[ids]
        if self.world == 1:
            return rows
        mine = self.owner(ids) == self.rank       # rows of other ranks are undefined here (may hold NaN): select, never scale
        rows = torch.where(mine.view(-1, *([1] * (rows.dim() - 1))), rows, torch.zeros((), dtype=rows.dtype, device=rows.device))
        if self.dry:
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

    def allgather_rows(self, table, group=None):
        """table [N, ...]: every rank has written the rows it owns; fills in everybody else's (in place)."""
        if self.world == 1:
            return table
        dev = table.device
        key = str(dev)
        if key not in self._gather_plan:
            rows = [self.rows_of(r, dev) for r in range(self.world)]
            m = max(int(x.numel()) for x in rows)
            dest = torch.cat([x for r, x in enumerate(rows) if r != self.rank])
            src = torch.cat([r * m + torch.arange(x.numel(), device=dev) for r, x in enumerate(rows) if r != self.rank])
            self._gather_plan[key] = (m, dest, src)
        m, dest, src = self._gather_plan[key]
        flat = table.reshape(self.num_nodes, -1)
        buf = torch.zeros((self.world * m, flat.shape[1]), dtype=table.dtype, device=dev)
        own = self.owned_rows(dev)
        buf[self.rank * m:self.rank * m + own.numel()] = flat[own]
        self._all_gather_blocks(buf, m, group)
        flat[dest] = buf[src]
        return table
