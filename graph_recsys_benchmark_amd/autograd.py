"""loss.backward() through the HIP conv stack (reference training step solvers.py:213-216).

    PEAStackFunction.apply(engine, x, n_slots, options, *conv parameters) -> stack [N, P, R]
                                      (the per-metapath representations that models/base.py:193-196 concatenates)

forward  = pea_model_forward_train (all P x S conv layers, softmax statistics kept in the workspace)
backward = per level, last to first:
             pea_model_backward_level  -- the sparse half in HIP: relu masks, bias / attention-vector gradient
                                          reductions, gradient gathers over the reversed relations (csrc/agg_bwd.hip)
             pea_grad_weight / pea_dense_batch on workspace views -- the dense half: dW = In^T dT (row-part MFMA
                                          reduction, fixed order), dIn = dT W and the first layer's dx = dT_0 W_cat (one
                                          deep-K job) on the forward transform kernels: no library GEMM in the step
The fusion and the BPR scorer on top of `stack` are differentiated by torch autograd on the batch's rows only
(models/base.py here: `_loss_autograd`), which is also what lets the last layer's gradient gathers skip every row outside
the batch (StackOptions.read_ids).
"""
import ctypes as C
import os

import torch

from . import _lib
from .engine import PARAM_SLOTS, RowSet, block_sum, dense_batch, grad_weight, mlp2_backward_data, mlp2_backward_data_sage


def _sparse_backward():
    import os
    return os.environ.get('PEA_SPARSE_BWD', '1') != '0'


class _Layout:
    """Parsed pea_model_describe output (float offsets from the 256-byte aligned workspace base)."""

    def __init__(self, engine):
        lib = _lib.load()
        need = C.c_int()
        _lib.check(lib.pea_model_describe(engine._h, None, 0, C.byref(need)))
        buf = (C.c_int64 * need.value)()
        _lib.check(lib.pea_model_describe(engine._h, buf, need.value, C.byref(need)))
        v = list(buf)
        self.n_levels, self.ld_x, self.off_x, self.off_dx, self.off_gpack, self.pack_floats, two_step = v[:7]
        self.two_step_train = bool(two_step)      # csrc/model.h: pea_model::fused2_train
        i = 7
        self.levels = []
        for _ in range(self.n_levels):
            names = ('ld_t', 'ld_o', 'off_t', 'off_o', 'off_dt', 'off_do', 'off_side', 'bias_off', 'att_src_off',
                     'att_dst_off', 'off_dad', 'off_das', 'ld_k', 'n_units')
            lv = dict(zip(names, v[i:i + 14]))
            i += 14
            units = []
            for _ in range(lv['n_units']):
                un = ('p', 's', 'rel', 'in_w', 'heads', 'F', 'HF', 'last', 'in_col', 't_col', 'o_col', 'b_off', 'ldb',
                      'bias_off')
                units.append(dict(zip(un, v[i:i + 14])))
                i += 14
            lv['units'] = units
            self.levels.append(lv)


def _view(wsf, off, n, ld):
    return wsf[off:off + n * ld].view(n, ld)


class _Slice:
    """Placeholder for a gradient that lives in the packed reduction buffer (materialised once at the end)."""

    def __init__(self, off, n, shape=None):
        self.off, self.n, self.shape = off, n, shape


def backward_conv_stack(engine, d_stack, x, layer_params, active_ids=None, compact=None, batch_flags=None):
    """Gradients of sum(stack * d_stack) wrt x and every conv parameter.  Returns (dx, [tuple per layer]).
    compact = (ids, rows): d_stack given as the gradient rows [len(ids), P * R] of the stack rows `ids` (int64; duplicates
    are summed in position order, ids < 0 skipped) instead of a dense [N, P, R] tensor (PEALossFunction).
    batch_flags (uint8 [N], optional): 1 on the rows of the stack that can carry a gradient."""
    lib = _lib.load()
    if not engine.enable_backward:
        raise RuntimeError('engine was built without enable_backward')
    lay = getattr(engine, '_layout', None)
    if lay is None:
        lay = engine._layout = _Layout(engine)
    kind, n, wsf = engine.kind, engine.plan.num_nodes, engine._wsf
    slots = PARAM_SLOTS[kind]
    # sharded plan: every rank handles the rows it owns; gradient rows the gathers over a reversed relation read from
    # other ranks are filled in between the two halves of a level (sharding.fill_in_*), the row-wise reductions give this
    # rank's SHARE of the parameter gradients (summed over the ranks at the end), dx comes out row-sharded (all-gathered)
    plan = engine.plan
    sharded = engine.sharded
    shard = plan.layout if sharded else None
    shard3 = plan.shard if sharded else None
    own32 = plan.own_rows_i32 if sharded else None
    to_reduce = []

    def rev_layout(rel):
        return plan.source_layouts[plan.reverse_of[rel]]
    first = [0]
    for s_ in engine.steps:
        first.append(first[-1] + s_)
    grads = [[None] * len(slots) for _ in range(engine.n_layers)]
    dx = None                       # written whole by the level-0 input-gradient job(s)
    gpack = wsf[lay.off_gpack:lay.off_gpack + lay.pack_floats]
    # 1. gradient of the last-layer outputs, internal column order
    dX = _view(wsf, lay.off_dx, n, lay.ld_x)
    if compact is not None:
        # the batch's gradient rows go straight into the internal column order, duplicates summed in a fixed order, one launch
        from .engine import rows_scatter_sum
        col_of = [0] * engine.P
        for lv in lay.levels:
            for u in lv['units']:
                if u['last']:
                    col_of[u['p']] = u['o_col']
        dX.zero_()
        rows_scatter_sum(compact[0], compact[1], engine.P, engine.repr_dim, col_of, dX)
    elif active_ids is not None:
        # d_stack is zero outside the rows the loss read: clear once, then move those rows only
        dX.zero_()
        rows = d_stack[active_ids]                                              # [B', P, R]
        for lv in lay.levels:
            for u in lv['units']:
                if u['last']:
                    dX[active_ids, u['o_col']:u['o_col'] + u['HF']] = rows[:, u['p'], :]
    else:
        for lv in lay.levels:
            for u in lv['units']:
                if u['last']:
                    dX[:, u['o_col']:u['o_col'] + u['HF']] = d_stack[:, u['p'], :]
    stream = _lib.current_stream()

    premasked = set()      # levels whose output gradients were written through a gated product (relu mask applied there)

    def level_call(level, phase):
        if level in premasked:
            phase |= _lib.BWD_PREMASKED
        _lib.check(lib.pea_model_backward_level(engine._h, level, phase, _lib.ptr(engine._ws), engine.workspace_bytes,
                                                stream))

    for s in range(lay.n_levels - 1, -1, -1):
        lv = lay.levels[s]
        T = _view(wsf, lv['off_t'], n, lv['ld_t'])
        dT = _view(wsf, lv['off_dt'], n, lv['ld_t'])
        dO = _view(wsf, lv['off_do'], n, max(lv['ld_o'], 4))
        prev = lay.levels[s - 1] if s > 0 else None
        In_all = x if s == 0 else _view(wsf, prev['off_o'], n, prev['ld_o'])
        dIn_all = None if s == 0 else _view(wsf, prev['off_do'], n, max(prev['ld_o'], 4))
        if kind == 'sage' and lay.two_step_train:
            # Two-step training schedule, SAGE (csrc/model_bwd.hip, csrc/mlp2_bwd.hip; forward: mlp2_sage_kernel).  Layer 2 ran
            # transform first: out = mean_j T_1[j] + R_1[i] with T_1 = H lin_rel1^T, R_1 = H lin_root1^T + bias1.
            units = lv['units']
            emb = x.shape[1]
            if s == 1:
                level_call(1, 0)     # d bias1 = colsum dX;  dT_1 = dX spread over the reversed relations (1 / deg_i each)
                ncol = units[-1]['t_col'] + units[-1]['HF']
                live = None
                if _sparse_backward():
                    # the rows whose hidden row received a gradient: in-neighbours of the batch's rows (dT_1) and, SAGE having
                    # no self loops, the batch's rows themselves (the root term: dX)
                    sets = getattr(engine, '_live_sets', None)
                    if sets is None:
                        sets = engine._live_sets = [RowSet(n, x.device), RowSet(n, x.device), 0, False]
                    if batch_flags is None:
                        aux = getattr(engine, '_aux_set', None)
                        if aux is None:
                            aux = engine._aux_set = RowSet(n, x.device)
                        batch_flags = aux.fill_from(dX, ncol).flags
                    sets[2] ^= 1
                    live = sets[sets[2]].fill_from(dT, ncol, also=batch_flags)
                engine._live_rows = live
                pairs = []
                for u in units:
                    In = In_all[:, u['in_col']:u['in_col'] + u['in_w']]                    # the channel's hidden rows H
                    pairs += [(dT[:, u['t_col']:u['t_col'] + u['HF']], In), (dX[:, u['o_col']:u['o_col'] + u['HF']], In)]
                dWs = grad_weight(pairs, rows=live)
                for q, u in enumerate(units):
                    li = first[u['p']] + u['s']
                    grads[li] = [dWs[2 * q], _Slice(lv['bias_off'] + u['t_col'], u['HF']), dWs[2 * q + 1]]
                continue
            nxt = lay.levels[1]
            dT1 = _view(wsf, nxt['off_dt'], n, nxt['ld_t'])
            H = _view(wsf, lv['off_o'], n, lv['ld_o'])
            side = _view(wsf, lv['off_side'], n, lv['ld_t'])                               # the root term's gradient blocks
            u1_of = {u1['p']: u1 for u1 in nxt['units']}
            live = getattr(engine, '_live_rows', None)
            if live is not None:
                # invariant (as for GAT / GCN below): dM_0 and the root blocks are zero outside this step's list, so the
                # reverse aggregation needs no per-row test
                sets = engine._live_sets
                if not sets[3]:
                    dT.zero_()
                    side.zero_()
                    sets[3] = True
                else:
                    sets[1 - sets[2]].zero_rows_of(dT, len(units) * emb)
                    sets[1 - sets[2]].zero_rows_of(side, len(units) * emb)
            elif getattr(engine, '_live_sets', None) is not None:
                engine._live_sets[3] = False
            zeros = getattr(engine, '_zeros_n', None)
            if zeros is None:
                zeros = engine._zeros_n = torch.zeros(n, dtype=torch.float32, device=x.device)
            chans, pairs = [], []
            n_rel, prev_rel = 0, None
            for u in units:
                li = first[u['p']] + u['s']
                c, u1 = u['t_col'], u1_of[u['p']]
                if u['rel'] != prev_rel:           # one mean per distinct first relation, shared by its channels (model.hip)
                    n_rel, prev_rel = n_rel + 1, u['rel']
                a0 = (n_rel - 1) * emb
                w_rel0, _b0, w_root0 = layer_params[li]
                w_rel1, _b1, w_root1 = layer_params[li + 1]
                chans.append((w_rel0, w_root0, w_rel1, w_root1, u1['t_col'], u1['o_col'], u['o_col'], c, c, c))
                # d lin_rel0 = dZ_0^T M_0 (the mean of a row without incoming edges is 0: its stale A_0 row is swapped for
                # 0 * x), d lin_root0 = dZ_0^T x
                pairs += [(dO[:, c:c + u['HF']], T[:, a0:a0 + emb], plan.edgeless_mask(u['rel']), x, zeros),
                          (dO[:, c:c + u['HF']], x)]
            mlp2_backward_data_sage(chans, emb, units[0]['HF'], u1_of[units[0]['p']]['HF'], dT1, dX, H, dO, dT, side, rows=live)
            dWs = grad_weight(pairs, rows=live)
            _lib.check(lib.pea_model_set_active_rows0(engine._h, None,
                                                      None if live is None else _lib.ptr(live.ids),
                                                      None if live is None else _lib.ptr(live.count)))
            level_call(0, 0)          # d bias0; per channel: dM_0 spread over the reversed relation + the root block -> over A_0
            dx = block_sum(T, len(units), emb)
            for q, u in enumerate(units):
                li = first[u['p']] + u['s']
                grads[li] = [dWs[2 * q], _Slice(lv['bias_off'] + u['t_col'], u['HF']), dWs[2 * q + 1]]
            continue
        if kind == 'sage':
            level_call(s, 0)
            dT.zero_()
            units = lv['units']
            pairs, jobs, root_jobs = [], [], []
            shared = {}
            for u in units:
                shared[u['t_col']] = shared.get(u['t_col'], 0) + 1
            for u in units:
                li = first[u['p']] + u['s']
                w_rel, _b, w_root = layer_params[li]
                G = (dX if u['last'] else dO)[:, u['o_col']:u['o_col'] + u['HF']]
                M = T[:, u['t_col']:u['t_col'] + u['in_w']]
                In = In_all[:, u['in_col']:u['in_col'] + u['in_w']]
                pairs += [(G, M), (G, In)]
                grads[li][1] = _Slice(u['bias_off'], u['HF'])
                if shared[u['t_col']] == 1:                                   # its own mean block: dM = G W_rel, written in place
                    jobs.append((G, w_rel, dT[:, u['t_col']:u['t_col'] + u['in_w']]))
                else:                                                         # level 0: channels of one relation share M
                    tmp = torch.empty((n, u['in_w']), dtype=torch.float32, device=x.device)
                    jobs.append((G, w_rel, tmp))
                    shared.setdefault('acc', []).append((u, tmp))
                if s > 0:
                    root_jobs.append((G, w_root, dIn_all[:, u['in_col']:u['in_col'] + u['in_w']]))
            dWs = grad_weight(pairs, shard=shard3)
            to_reduce.extend(dWs)
            for q, u in enumerate(units):
                li = first[u['p']] + u['s']
                grads[li][0], grads[li][2] = dWs[2 * q], dWs[2 * q + 1]
            if s == 0:
                # dx = sum_u G_u W_root_u (+ the mean-path gradient below): one deep-K job when the G blocks are the
                # contiguous columns of dO (every 2-step model), else one job per channel and a sum
                cont = sorted((u for u in units if not u['last']), key=lambda u: u['o_col'])
                ocols = sum(u['HF'] for u in cont)
                contiguous = len(cont) == len(units) and all(
                    cont[k]['o_col'] == sum(v['HF'] for v in cont[:k]) for k in range(len(cont)))
                dx = torch.empty_like(x)
                if contiguous:
                    w_cat = torch.cat([layer_params[first[u['p']] + u['s']][2] for u in cont], dim=0)   # [ocols, emb]
                    root_jobs.append((dO[:, :ocols], w_cat, dx))
                else:
                    parts = torch.empty((len(units),) + tuple(x.shape), dtype=torch.float32, device=x.device)
                    for q, u in enumerate(units):
                        G = (dX if u['last'] else dO)[:, u['o_col']:u['o_col'] + u['HF']]
                        root_jobs.append((G, layer_params[first[u['p']] + u['s']][2], parts[q]))
            dense_batch(jobs + root_jobs, rows=own32)
            if s == 0 and not contiguous:
                torch.sum(parts, dim=0, out=dx)
            for u, tmp in shared.get('acc', []):
                dT[:, u['t_col']:u['t_col'] + u['in_w']] += tmp
            if sharded:     # the reverse mean aggregation gathers dM rows of the forward relation's destinations
                seen, items = set(), []
                for u in units:
                    if u['t_col'] not in seen:
                        seen.add(u['t_col'])
                        items.append((dT, u['t_col'], u['in_w'], rev_layout(u['rel'])))
                shard.fill_in_rows_batch(items)         # one exchange for the level
            level_call(s, 1)
            dagg = _view(wsf, lv['off_side'], n, lv['ld_t'])
            done = set()
            for u in units:
                if s == 0:
                    if u['t_col'] not in done:
                        dx += dagg[:, u['t_col']:u['t_col'] + u['in_w']]
                        done.add(u['t_col'])
                else:
                    dIn_all[:, u['in_col']:u['in_col'] + u['in_w']] += dagg[:, u['t_col']:u['t_col'] + u['in_w']]
            continue
        units = lv['units']
        if s == 0 and lay.two_step_train:
            # Two-step training schedule (csrc/model.h: fused2_train; GAT with one head or GCN, single GPU; SAGE: above).  dO_0 holds dZ_0, the
            # gradient of the first transform's pre-activations (masked by the gated product above); A_0 (T_0 region) holds the
            # aggregates of the rows with incoming edges (the others' input is x itself).  Dense half on views: dW_0 = dZ_0^T A_0 and dA_0 = dZ_0 W_0 per channel;
            # then ONE call runs the softmax passes in x space (bias gradient, D pass, S pass -> per-channel dx parts over A_0).
            emb = x.shape[1]
            # data path of the dense half in ONE launch (csrc/mlp2_bwd.hip): dZ_0 = (dT_1 W_1) masked by H > 0 -> dO_0 region,
            # dA_0 = dZ_0 W_0 -> dT_0 region, the hidden gradient tile staying in registers between the two products
            nxt = lay.levels[1]
            dT1 = _view(wsf, nxt['off_dt'], n, nxt['ld_t'])
            H = _view(wsf, lv['off_o'], n, lv['ld_o'])
            u1_of = {u1['p']: u1 for u1 in nxt['units']}
            chans, pairs, Ws = [], [], []
            gcn = kind == 'gcn'
            from_col = engine._gcn_from_col if gcn else False
            for u in units:
                li = first[u['p']] + u['s']
                c, u1 = u['t_col'], u1_of[u['p']]
                chans.append((layer_params[li][0], layer_params[li + 1][0], u1['t_col'], u['o_col'], c, c))
                # dW_0 = dZ_0^T In ([HF, emb] = GAT lin.weight's layout; GCN's weight is its transpose); In = A_0 where the node
                # has incoming edges, x where not (GCN: x times the self-loop norm deg^-1)
                pair = (dO[:, c:c + u['HF']], T[:, c:c + emb], plan.edgeless_mask(u['rel']), x)
                pairs.append(pair + (plan.gcn_self_norm(u['rel'], from_col),) if gcn else pair)
                Ws.append(layer_params[li][0])
            live = getattr(engine, '_live_rows', None) if _sparse_backward() else None
            if live is not None:
                # invariant: dA_0 (the dT_0 region) is zero on every row outside this step's list, so the gradient gathers
                # below need no per-edge test (3/4 of the gathered rows are live: a test costs more than it saves) -- the
                # whole region is cleared once, afterwards only the rows the previous step wrote
                sets = engine._live_sets
                if not sets[3]:
                    dT.zero_()
                    sets[3] = True
                else:
                    sets[1 - sets[2]].zero_rows_of(dT, len(units) * emb)
            elif getattr(engine, '_live_sets', None) is not None:
                engine._live_sets[3] = False         # a dense step writes every row: the invariant starts over
            mlp2_backward_data(chans, emb, units[0]['HF'], u1_of[units[0]['p']]['HF'], dT1, H, dO, dT, rows=live, weights_in_out=gcn)
            dWs = grad_weight(pairs, rows=live)
            # no flags: 3/4 of the rows these gathers fetch are live, and a test per edge costs more than the quarter of the
            # fetches it saves -- GAT S pass 1.12 -> 1.23 ms, GCN reverse aggregation 1.47 -> 1.75, SAGE 0.82 -> 1.09 on the
            # 25m-shaped graph (profiles/r03/bwd0_filter_r03.txt; the GCN / SAGE test lived in the forward kernels and was removed)
            _lib.check(lib.pea_model_set_active_rows0(engine._h, None,
                                                      None if live is None else _lib.ptr(live.ids),
                                                      None if live is None else _lib.ptr(live.count)))
            level_call(0, 0)
            n_ch = len(units)
            dx = block_sum(T, n_ch, emb)                                             # the S pass wrote the channels' parts over A_0
            if gcn:                      # no attention vectors: weight [in, out] = the transpose of the reduced block, bias
                for q, u in enumerate(units):
                    li = first[u['p']] + u['s']
                    grads[li][0] = dWs[q].t()
                    grads[li][1] = _Slice(lv['bias_off'] + u['t_col'], u['HF'])
                continue
            das = _view(wsf, lv['off_das'], n, lv['ld_k'])
            dad = _view(wsf, lv['off_dad'], n, lv['ld_k'])
            d_ws, d_wd = grad_weight([(das, x), (dad, x)])                           # [ld_k, emb]: rows = channels in unit order
            d_ws, d_wd = d_ws[:n_ch], d_wd[:n_ch]
            # the logits came from x . ws, x . wd with ws = W_0^T att_j, wd = W_0^T att_i: chain rule, batched over the channels
            W = torch.stack(Ws)                                                      # [P, HF, emb]
            att_i = torch.stack([layer_params[first[u['p']] + u['s']][1].reshape(-1) for u in units])   # [P, HF]
            att_j = torch.stack([layer_params[first[u['p']] + u['s']][2].reshape(-1) for u in units])
            d_att_j = torch.bmm(W, d_ws.unsqueeze(2)).squeeze(2)
            d_att_i = torch.bmm(W, d_wd.unsqueeze(2)).squeeze(2)
            dW0 = torch.stack(dWs)
            dW0.addcmul_(att_j.unsqueeze(2), d_ws.unsqueeze(1)).addcmul_(att_i.unsqueeze(2), d_wd.unsqueeze(1))
            for q, u in enumerate(units):
                li = first[u['p']] + u['s']
                shape = layer_params[li][1].shape
                grads[li][0] = dW0[q]
                grads[li][1] = d_att_i[q].view(shape)
                grads[li][2] = d_att_j[q].view(shape)
                grads[li][3] = _Slice(lv['bias_off'] + u['t_col'], u['HF'])
            continue
        level_call(s, 0)
        if sharded:
            # runs of adjacent channels on one relation: their output-gradient columns (and GAT side records) travel together
            side = _view(wsf, lv['off_side'], n, 4 * max(sum(u['heads'] for u in units), 1)) if kind == 'gat' else None
            runs, a_k = [], 0
            for u in units:
                r = runs[-1] if runs else None
                if r and r['rel'] == u['rel'] and r['last'] == u['last'] and r['col'] + r['w'] == u['o_col']:
                    r['w'] += u['HF']
                    r['heads'] += u['heads']
                else:
                    runs.append(dict(rel=u['rel'], last=u['last'], col=u['o_col'], w=u['HF'], a_k=a_k, heads=u['heads']))
                a_k += u['heads']
            # rows every rank owns are complete rows: the batch's rows of the last level travel whole (one all-reduce per
            # buffer instead of one per run of channels), the other levels' runs in one batched exchange
            batch_items, ids_done = [], False
            for r in runs:
                G = dX if r['last'] else dO
                if r['last'] and active_ids is not None:       # only the batch's rows carry a gradient at the last layer
                    if not ids_done:
                        shard.fill_in_ids(dX, 0, dX.shape[1], active_ids)
                        if side is not None:
                            shard.fill_in_ids(side, 0, side.shape[1], active_ids)
                        ids_done = True
                else:
                    batch_items.append((G, r['col'], r['w'], rev_layout(r['rel'])))
                    if side is not None:
                        batch_items.append((side, 4 * r['a_k'], 4 * r['heads'], rev_layout(r['rel'])))
            shard.fill_in_rows_batch(batch_items)
            level_call(s, 2)
        if s == 0:
            # every first-layer channel reads x: one GEMM for all weight gradients, one for dx
            ncol = units[-1]['t_col'] + units[-1]['HF']
            dT0 = dT[:, :ncol]
            dx = torch.empty_like(x)
            if kind == 'gat':
                dW_all = grad_weight([(dT0, x)], shard=shard3)[0]           # [sum HF, emb]
                w_cat = torch.cat([layer_params[first[u['p']] + u['s']][0] for u in units], dim=0)
            else:
                dW_all = grad_weight([(x, dT0)], shard=shard3)[0]           # [emb, sum F]
                w_cat = torch.cat([layer_params[first[u['p']] + u['s']][0] for u in units], dim=1).t().contiguous()
            to_reduce.append(dW_all)
            dense_batch([(dT0, w_cat, dx)], rows=own32)                     # dx = dT_0 W_cat: one deep-K job (K = sum HF)
        # weight gradients of the level in one launch pair, input gradients in one launch (dense_bwd.hip)
        if s > 0:
            # dIn = dT W is the output gradient of the level below BEFORE its relu mask; In (= that level's relu output)
            # is the mask: applied in the product's epilogue when every job qualifies (k <= 128, more than 16 outputs:
            # the persistent transform kernel), so that level's own mask pass over the buffer is skipped
            gated = all(u['HF'] <= 128 and u['in_w'] > 16 for u in units)
            pairs, dense = [], []
            for u in units:
                li = first[u['p']] + u['s']
                dTu = dT[:, u['t_col']:u['t_col'] + u['HF']]
                In = In_all[:, u['in_col']:u['in_col'] + u['in_w']]
                dIn = dIn_all[:, u['in_col']:u['in_col'] + u['in_w']]
                if kind == 'gat':
                    pairs.append((dTu, In))                                     # [HF, in]
                    dense.append((dTu, layer_params[li][0], dIn) + ((In,) if gated else ()))               # dT @ W
                else:
                    pairs.append((In, dTu))                                     # [in, F]
                    dense.append((dTu, layer_params[li][0].t().contiguous(), dIn) + ((In,) if gated else ()))
            live = None
            if lay.two_step_train and s == 1 and _sparse_backward():
                # Gradient support: the loss read the batch's rows only, so dT_1 is identically zero outside the batch rows and
                # their layer-2 in-neighbours (about 1/4 of the nodes on the 25m-shaped graph).  The dense half of both layers
                # and the first layer's gradient gathers walk that row set (built on the device, count never read by the host).
                sets = getattr(engine, '_live_sets', None)
                if sets is None:      # two sets: this step's and the previous step's (its rows of dA_0 are zeroed again below)
                    sets = engine._live_sets = [RowSet(n, x.device), RowSet(n, x.device), 0, False]
                sets[2] ^= 1
                live = sets[sets[2]]
                live.fill_from(dT, units[-1]['t_col'] + units[-1]['HF'])      # all channels' columns of dT_1
                engine._live_rows = live
            dWs = grad_weight(pairs, shard=shard3, rows=live)
            to_reduce.extend(dWs)
            if not (lay.two_step_train and s == 1):      # two-step training: the level-0 branch runs both products fused
                dense_batch(dense, rows=own32)
            if gated:
                premasked.add(s - 1)
        for q, u in enumerate(units):
            li = first[u['p']] + u['s']
            if kind == 'gat':
                grads[li][0] = dW_all[u['t_col']:u['t_col'] + u['HF']] if s == 0 else dWs[q]
                shape = layer_params[li][1].shape
                grads[li][1] = _Slice(lv['att_dst_off'] + u['t_col'], u['HF'], shape)
                grads[li][2] = _Slice(lv['att_src_off'] + u['t_col'], u['HF'], shape)
                grads[li][3] = _Slice(lv['bias_off'] + u['t_col'], u['HF'])
            else:
                grads[li][0] = dW_all[:, u['t_col']:u['t_col'] + u['HF']] if s == 0 else dWs[q]
                grads[li][1] = _Slice(lv['bias_off'] + u['t_col'], u['HF'])
    # the small (bias / attention-vector) gradients were reduced into the packed buffer level by level: one copy of it,
    # then views (the workspace itself is overwritten by the next step)
    packed = gpack.clone()
    if sharded:
        shard.all_reduce_sum_(to_reduce + [packed])      # the ranks' shares of every parameter gradient, one collective
        shard.allgather_rows(dx)                         # dx rows are complete on their owners
    for g in grads:
        for q, item in enumerate(g):
            if isinstance(item, _Slice):
                t = packed[item.off:item.off + item.n]
                g[q] = t if item.shape is None else t.view(item.shape)
    return dx, grads


class StackOptions:
    """Per-call options of PEAStackFunction (a plain object: autograd passes it through untouched).
    fuse_att / fuse_masked: what the by-product fused table (`fused`, set by the forward) is fused with (None: zeros).
    read_ids: the only rows of the returned stack the caller will read (int64 ids, duplicates allowed): only they can
    carry a gradient, so the backward moves just those rows of d_stack and (GAT) lets the last layer's gradient gathers
    skip every other row."""

    def __init__(self, fuse_att=None, fuse_masked=None, read_ids=None):
        self.fuse_att, self.fuse_masked, self.read_ids = fuse_att, fuse_masked, read_ids
        self.fused = None


class PEAStackFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, x, n_slots, options, *flat):
        layer_params = [tuple(flat[i:i + n_slots]) for i in range(0, len(flat), n_slots)]
        # the fused table of the same launch is a free by-product (not differentiated here: the caller fuses the rows
        # it needs with torch ops on the stack)
        options = options or StackOptions()
        att = options.fuse_att
        if att is None:
            att = torch.zeros(engine.P, engine.repr_dim, device=x.device)
        # sharded: each rank keeps the rows it owns (fused table and stack are defined there only); the caller exchanges
        # the rows it reads (models/base.py: _loss_autograd)
        options.fused, stack = engine.forward(layer_params, x, att=att, masked=options.fuse_masked, want_stack=True,
                                              train=True, gather=False)
        ctx.engine, ctx.n_slots = engine, n_slots
        ctx.active_ids, ctx.active_rows = options.read_ids, None
        if options.read_ids is not None:      # (every kind: the last layer's gradient gathers skip the rows not flagged)
            ctx.active_rows = torch.zeros(x.shape[0], dtype=torch.uint8, device=x.device)
            ctx.active_rows.index_fill_(0, options.read_ids, 1)      # (indexed assignment of a Python scalar stages it through the host)
        ctx.save_for_backward(x, *[t for t in flat if t is not None])
        ctx.present = [t is not None for t in flat]
        return stack

    @staticmethod
    def backward(ctx, d_stack):
        saved = list(ctx.saved_tensors)
        x, rest = saved[0], saved[1:]
        flat, it = [], iter(rest)
        for p in ctx.present:
            flat.append(next(it) if p else None)
        n_slots = ctx.n_slots
        layer_params = [tuple(flat[i:i + n_slots]) for i in range(0, len(flat), n_slots)]
        lib = _lib.load()
        mask = ctx.active_rows        # uint8 [N] or None: rows of the final outputs that can carry a gradient
        with torch.no_grad():
            if mask is not None:
                _lib.check(lib.pea_model_set_active_rows(ctx.engine._h, _lib.ptr(mask)))
            try:
                dx, grads = backward_conv_stack(ctx.engine, d_stack.contiguous(), x, layer_params, ctx.active_ids)
            finally:
                if mask is not None:
                    _lib.check(lib.pea_model_set_active_rows(ctx.engine._h, None))
        out = []
        for lp, g in zip(layer_params, grads):
            for t, gt in zip(lp, g):
                out.append(None if t is None else gt.reshape(t.shape))
        return (None, dx, None, None, *out)


class PEALossFunction(torch.autograd.Function):
    """The whole training-step loss of a PEA model as ONE autograd node: conv stack forward (HIP), the batch's stack rows,
    fusion + scorer + BPR loss with their backward (csrc/bpr_train.hip), and in backward() the batch's gradient rows
    scattered straight into the output-gradient buffer (pea_rows_scatter_sum) before the conv stack's HIP backward.
    Compared with PEAStackFunction + torch ops on top, autograd never builds the dense [N, P, R] gradient of the stack
    (two 158 MB fills, a sort-based index backward and nine index_puts per step on the 25m-shaped graph).
    Reference: solvers.py:213-214 (loss = model.loss(batch); loss.backward())."""

    @staticmethod
    def forward(ctx, engine, x, n_slots, options, ids, att, fc1_w, fc1_b, fc2_w, fc2_b, *flat):
        from .engine import bpr_train_raw
        layer_params = [tuple(flat[i:i + n_slots]) for i in range(0, len(flat), n_slots)]
        options = options or StackOptions()
        fuse_att = options.fuse_att
        if fuse_att is None:
            fuse_att = torch.zeros(engine.P, engine.repr_dim, device=x.device)
        # The loss reads the batch's stack rows only: they are picked from the workspace's X region (the last layer's outputs in
        # the schedule's own column order) instead of having the fusion launch write the whole [N, P, R] stack for them
        # (158 MB on the 25m-shaped graph: 0.083 -> 0.045 ms for that launch)
        options.fused = engine.forward(layer_params, x, att=fuse_att, masked=options.fuse_masked, want_stack=False,
                                       train=True, gather=False)
        lay = getattr(engine, '_layout', None)
        if lay is None:
            lay = engine._layout = _Layout(engine)
        cols = getattr(engine, '_stack_cols', None)
        if cols is None:
            col_of = [0] * engine.P
            for lv in lay.levels:
                for u in lv['units']:
                    if u['last']:
                        col_of[u['p']] = u['o_col']
            cols = torch.tensor([c + r for c in col_of for r in range(engine.repr_dim)], dtype=torch.int64, device=x.device)
            engine._stack_cols = cols
        table = _view(engine._wsf, lay.off_x, x.shape[0], lay.ld_x)
        if engine.sharded:
            # every rank holds the stack rows it owns: the batch's rows are summed from their owners (one all-reduce of
            # [3B, P * R], exact: x + 0); the head is computed replicated; only owned rows receive a gradient here
            layout = engine.plan.layout
            picked = layout.gather_rows(table, ids).index_select(1, cols)
            ids_b = torch.where(layout.owner(ids) == layout.rank, ids, torch.full_like(ids, -1))
        else:
            picked, ids_b = table.index_select(0, ids).index_select(1, cols), ids
        loss, grad_rows, head = bpr_train_raw(picked.view(-1, engine.P, engine.repr_dim), att, fc1_w, fc1_b, fc2_w, fc2_b)
        ctx.engine, ctx.n_slots, ctx.ids, ctx.ids_b = engine, n_slots, ids, ids_b
        ctx.grad_rows, ctx.head = grad_rows, head
        ctx.active_rows = None
        # the batch's rows: the only rows of the last layer's output gradient that are non-zero (GAT: D / S passes; GCN and
        # SAGE: the reverse aggregation does not fetch the others; two-step SAGE: part of the gradient's support)
        ctx.active_rows = torch.zeros(x.shape[0], dtype=torch.uint8, device=x.device)
        ctx.active_rows.index_fill_(0, ids, 1)
        ctx.save_for_backward(x, *[t for t in flat if t is not None])
        ctx.present = [t is not None for t in flat]
        return loss

    @staticmethod
    def backward(ctx, g):
        saved = list(ctx.saved_tensors)
        x, rest = saved[0], saved[1:]
        flat, it = [], iter(rest)
        for p in ctx.present:
            flat.append(next(it) if p else None)
        n_slots = ctx.n_slots
        layer_params = [tuple(flat[i:i + n_slots]) for i in range(0, len(flat), n_slots)]
        lib = _lib.load()
        mask = ctx.active_rows
        with torch.no_grad():
            if mask is not None:
                _lib.check(lib.pea_model_set_active_rows(ctx.engine._h, _lib.ptr(mask)))
            # the conv stack's backward is linear in the batch's gradient rows: the upstream gradient scales them once
            # ([3B, P * R]) instead of every one of the ~80 parameter gradients afterwards
            try:
                dx, grads = backward_conv_stack(ctx.engine, None, x, layer_params, ctx.ids, compact=(ctx.ids_b, ctx.grad_rows * g),
                                                batch_flags=mask)
            finally:
                if mask is not None:
                    _lib.check(lib.pea_model_set_active_rows(ctx.engine._h, None))
            out = []
            for lp, gl in zip(layer_params, grads):
                for t, gt in zip(lp, gl):
                    out.append(None if t is None else gt.reshape(t.shape))
            head = tuple(None if t is None else t * g for t in ctx.head)
            return (None, dx, None, None, None) + head + tuple(out)
