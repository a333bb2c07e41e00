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
