"""Data-parallel path on CPU with gloo, world_size 2 (the driver runs the real
RCCL job): bucketed all-reduce of the flat gradient buffer, global token-count
normalisation, batch sharding.  The per-rank gradients come from the oracle's
autograd so the identity checked is the one the GPU trainer relies on:

    sum_r  d/dtheta [ sum_local_r(xent) / sum_global(mask) ]  ==  full-batch gradient.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    torch.set_num_threads(1)
    from neuralmonkey_amd import distributed, synthetic
    from neuralmonkey_amd.variables import VariableStore, random_normal_initializer
    from oracle import nm_oracle as O
    from oracle import torch_ref as TR

    dp = distributed.init_from_env(backend="gloo")
    assert dp.rank == rank and dp.world_size == world
    dp.sharded = False          # this test: the replicated update's exchange (spans, early spans, rows); the sharded
                                # optimizer has test_sharded_optimizer_equals_the_replicated_update below

    vocab, dim, batch = 50, 8, 6
    params = O.init_params(seed=5, vocab_src=vocab, vocab_tgt=vocab, emb=dim, rnn=dim, std=0.2)
    ds = synthetic.synthetic_dataset(seed=6, batch=batch, src_len=7, tgt_len=6, vocab=vocab, ragged=True)
    shard = dp.shard(ds)
    assert len(shard) == batch // world

    def arrays(d):
        src = O.pad_ids([list(s) for s in d.get_series("source")], 7)
        tgt = O.pad_ids([list(s) for s in d.get_series("target")], 7, add_end_symbol=True)
        return src, np.ascontiguousarray(tgt.T)

    src, tgt = arrays(shard)
    local_count = float((tgt != 0).sum())
    global_count = dp.all_reduce_scalar(local_count)

    # rank-local gradient of sum_local(xent)/sum_global(mask), no regulariser (added once, after the reduce)
    tp = TR.to_torch(params)
    loss = TR.train_forward(tp, src, tgt) * (local_count / global_count)
    loss.backward()

    store = VariableStore("cpu", seed=0)
    for name, val in params.items():
        store.declare(name, val.shape if val.shape else (1,), random_normal_initializer())
    store.finalize()
    store.ensure_grad()
    for name in params:
        store.g(name).copy_(tp[name].grad.reshape(store.g(name).shape))
    dp.bucket_elems = 1000                      # force several buckets
    # two slices start their reduction early (what Decoder.backward does for the vocabulary projection and the
    # decoder embeddings while the rest of the backward pass runs); the final call covers the remaining spans
    names = sorted(params)
    early = [names[len(names) // 2], names[0]]
    dp.begin_step()
    dp.all_reduce_early(store, [early[0]])
    dp.all_reduce_early(store, [early[1]])
    try:
        dp.all_reduce_early(store, [early[0]])
        raise AssertionError("a span was reduced twice")
    except RuntimeError:
        pass
    dp.all_reduce_gradients(store)
    assert not dp._early and not dp._handles

    # ---- the same step with the encoder embeddings exchanged as (ids, rows) (NM_DP_SPARSE_EMB=1): the device
    # primitives are stand-ins here (the product path passes libnmhip kernels, tests/test_dp_gpu.py runs those);
    # what is under test is the protocol -- counts, padding, rank order, span bookkeeping
    dense = {name: store.g(name).clone() for name in params}
    for name in params:
        store.g(name).copy_(tp[name].grad.reshape(store.g(name).shape))
    emb_name = "encoder_input/embedding_matrix_0"
    calls = []

    def gather_rows(src_t, idx, dst):
        dst.copy_(src_t[idx.long()])

    def scatter_add(table, ids, rows):
        assert len(set(ids.tolist())) == len(ids), "an id twice within one launch"
        calls.append(len(ids))
        table.index_add_(0, ids.long(), rows)

    dp.sparse_embeddings = True
    dp.begin_step()
    assert dp.exchange_sparse_rows(store, emb_name, src, gather_rows, scatter_add,
                                   lambda a, b: b.copy_(-a))
    assert len(calls) == 1 + world                       # cancel the local rows, then one launch per rank
    try:
        dp.all_reduce_early(store, [emb_name])
        raise AssertionError("a span was reduced twice")
    except RuntimeError:
        pass
    dp.all_reduce_gradients(store)
    dp.sparse_embeddings = False
    for name in params:
        a, b = store.g(name), dense[name]
        assert float((a - b).abs().max()) <= 1e-6 * max(float(b.abs().max()), 1e-6), name
    np.save(os.path.join(out_dir, "emb_rank{}.npy".format(rank)), store.g(emb_name).numpy())

    if rank == 0:
        fsrc, ftgt = arrays(ds)
        ftp = TR.to_torch(params)
        TR.train_forward(ftp, fsrc, ftgt).backward()
        worst = 0.0
        for name in params:
            want = ftp[name].grad.reshape(-1).numpy()
            got = store.g(name).reshape(-1).numpy()
            worst = max(worst, float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-5)))
        np.save(os.path.join(out_dir, "result.npy"), np.array([worst, global_count, float((ftgt != 0).sum())]))
    # what a bench line reports about the exchange (bench.py "dp", tools/scale.sh): every rank answered an all-gather,
    # bytes per step = the flat gradient buffer, the early spans counted apart
    assert dp.ranks_seen() == world
    report = dp.exchange_report()
    assert report["ranks_seen"] == world and report["bytes"] == store.total * 4
    assert 0 <= report["early_bytes"] <= report["bytes"]
    # parameter broadcast makes replicas identical
    store.theta.add_(float(rank))
    dp.broadcast_parameters(store, src=0)
    assert float(store.theta.sum()) == pytest.approx(float(store.theta.sum()))
    distributed.shutdown()


def test_sharded_gradients_equal_full_batch(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    worst, global_count, full_count = np.load(tmp_path / "result.npy")
    assert global_count == full_count
    assert worst < 5e-4, worst          # fp32 re-association between shard and full-batch sums
    # sparse exchange: every rank added the same blocks in the same order -> bit-identical replicas
    assert np.array_equal(np.load(tmp_path / "emb_rank0.npy"), np.load(tmp_path / "emb_rank1.npy"))


def test_single_process_is_a_no_op(monkeypatch):
    from neuralmonkey_amd import distributed
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert distributed.init_from_env() is None and distributed.current() is None


def test_shard_sizes_are_even_and_tiny_batches_are_refused():
    """3 ranks, 8 rows -> 3/3/2 contiguous rows covering the batch; fewer rows than ranks raise on every rank
    alike (an empty shard would leave its rank out of the step while the others wait in all_reduce)."""
    from neuralmonkey_amd.dataset import BatchingScheme, Dataset
    from neuralmonkey_amd.distributed import DataParallel
    ds = Dataset("d", {"source": [[str(i)] for i in range(8)]}, BatchingScheme(batch_size=8))
    seen = []
    for rank in range(3):
        dp = DataParallel.__new__(DataParallel)
        dp.rank, dp.world_size = rank, 3
        part = dp.shard(ds)
        seen.append([s[0] for s in part.get_series("source")])
    assert [len(p) for p in seen] == [3, 3, 2]
    assert sum(seen, []) == [str(i) for i in range(8)]
    dp.world_size = 9
    with pytest.raises(ValueError, match="cannot be sharded"):
        dp.shard(ds)


# ---- sharded optimizer: reduce-scatter -> update of the rank's slices -> all-gather -------------------
def _sharded_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), NM_DP_BIG_VARIABLE="3000")
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    torch.set_num_threads(1)
    import torch.distributed as tdist
    from neuralmonkey_amd import distributed
    from neuralmonkey_amd.variables import VariableStore, random_normal_initializer
    from tests.cpu_optim_tables import CpuOptimizerTables

    dp = distributed.init_from_env(backend="gloo")
    dp.bucket_elems = 2501                          # several buckets per big variable, none divisible by 8
    shapes = {"enc/embedding": (40, 101), "enc/kernel": (37, 9), "enc/bias": (9,), "dec/logits/W": (11, 613),
              "dec/logits/bias": (613,), "dec/small": (5,), "dec/embedding": (77, 53)}

    def fresh_store():
        st = VariableStore("cpu", seed=11)
        for name, shape in shapes.items():
            st.declare(name, shape, random_normal_initializer(stddev=0.3))
        st.finalize()
        st.ensure_grad()
        st.ensure_adam()
        return st

    plan = dp.plan(fresh_store())
    assert any(name for _, _, _, name in plan.buckets) and any(name is None for _, _, _, name in plan.buckets)
    assert plan.tails(), "the shapes were chosen so that buckets leave indivisible tails"
    assert len(plan.of_variable["dec/logits/W"]) >= 2
    owned = sorted(plan.owned(0) + plan.owned(1) + plan.tails())
    assert owned[0][0] == 0 and owned[-1][1] == plan.total and all(a[1] == b[0] for a, b in zip(owned, owned[1:]))

    def run(sharded, early):
        st = fresh_store()
        dp.sharded, dp.poison_foreign = sharded, sharded
        regularizable = {n for n in shapes if "bias" not in n}
        tables = CpuOptimizerTables(st, regularizable, set(shapes), cuts=dp.optimizer_cuts(st))
        m, v = st.ensure_adam()
        word = torch.zeros(1, dtype=torch.int32)
        clipped = []
        thetas = []
        for step in range(1, 5):
            g = torch.Generator().manual_seed(100 * step + rank)
            grad = st.ensure_grad()
            grad.copy_(torch.randn(st.total, generator=g) * (0.02 if step != 2 else 0.4))
            word.fill_(1 if (step == 3 and rank == 1) else 0)       # one rank's time loop "gave up" in step 3
            dp.begin_step()
            if early:
                dp.all_reduce_early(st, ["dec/logits/W", "dec/logits/bias"])
            lr_t = 1e-3 * (1 - 0.999 ** step) ** 0.5 / (1 - 0.9 ** step)
            dp.optimizer_step(st, tables, 0, m, v, 1e-4, 1e-3, 1.0, (lr_t, 0.9, 0.999, 1e-8), skip=word)
            norms = torch.sqrt(tables.workspace[3 * tables.nchunk:])
            clipped.append(int((norms > 1.0).sum()))
            thetas.append(st.theta.clone())
            assert int(word.item()) == (1 if step == 3 else 0), "the error word is the maximum over ranks"
        dp.gather_optimizer_slots(st, m, v)
        return thetas, m.clone(), v.clone(), clipped, tables.l1l2.clone()

    ref, ref_m, ref_v, clipped, ref_l1l2 = run(sharded=False, early=False)
    assert 0 < clipped[1] and clipped[0] < len(shapes), "the clip must bite on some tensors and not on all"
    assert torch.equal(ref[1], ref[2]) and not torch.equal(ref[0], ref[1]), "step 3 is skipped on every rank"
    for early in (False, True):
        got, got_m, got_v, _, l1l2 = run(sharded=True, early=early)
        for a, b in zip(got, ref):
            assert torch.isfinite(a).all()
            assert torch.equal(a, b), "sharded update differs from the replicated one"
        assert torch.equal(got_m, ref_m) and torch.equal(got_v, ref_v) and torch.equal(l1l2, ref_l1l2)
    both = [torch.zeros_like(ref[-1]) for _ in range(world)]
    tdist.all_gather(both, got[-1])
    assert torch.equal(both[0], both[1]), "replicas differ"
    report = dp.exchange_report()
    assert report["optimizer"] == "sharded" and report["optimizer_elements_per_rank"] < 0.6 * plan.total
    if rank == 0:
        np.save(os.path.join(out_dir, "sharded_ok.npy"), np.array([1]))
    distributed.shutdown()


def test_sharded_optimizer_equals_the_replicated_update(tmp_path):
    """Reduce-scatter -> norms from the ranks' partial sums -> clip + Adam on the rank's slices -> all-gather gives,
    bit for bit, what every rank computes when all of them reduce and update everything (generic_trainer.py:179-195);
    what a rank does not own is never read (it is filled with NaN after the reduction); a device error word raised
    on one rank voids the step on all."""
    mp.spawn(_sharded_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "sharded_ok.npy").exists()
