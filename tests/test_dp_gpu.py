"""Data-parallel trainer on the device, two ranks on ONE GPU over gloo (NM_DIST_BACKEND=gloo; the
driver runs the real RCCL job on 8 GPUs): every rank runs the HIP training path on its shard of the
batch -- global token-count normalisation, early all-reduce of the vocabulary-projection and
decoder-embedding gradient slices from inside the backward pass, bucketed reduction of the rest,
clip + Adam on identical data -- and must end up where one process training on the whole batch
ends up (sum_r d/dtheta[sum_local_r(xent) / sum_global(mask)] == full-batch gradient)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

VOCAB, DIM, BATCH, LEN = 120, 16, 8, 9


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _model(device="cuda:0"):
    from neuralmonkey_amd import synthetic
    return synthetic.build_translation_model(vocab_src=VOCAB, vocab_tgt=VOCAB, emb=DIM, rnn=DIM, max_len=LEN,
                                             beam_size=0, l2_weight=1e-4, clip_norm=1.0, device=device, seed=21)


def _batch():
    from neuralmonkey_amd import synthetic
    return synthetic.synthetic_dataset(seed=5, batch=BATCH, src_len=LEN, tgt_len=LEN - 1, vocab=VOCAB, ragged=True)


def _worker(rank, world, port, out_dir, backend="gloo", sparse=False, sharded=True):
    """backend gloo: all ranks share GPU 0; backend nccl (= RCCL): rank r owns GPU r."""
    local = rank if backend == "nccl" else 0
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(local), NM_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0",
                      NM_DP_SPARSE_EMB="1" if sparse else "0", NM_DP_SHARDED="1" if sharded else "0",
                      NM_DP_BIG_VARIABLE="1500")             # (the toy model's matrices own their buckets)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from neuralmonkey_amd import distributed
    torch.cuda.set_device(local)
    dp = distributed.init_from_env()
    assert dp is not None and dp.world_size == world and dp.overlap and dist.get_backend() == backend
    assert dp.sparse_embeddings == sparse and dp.sharded_active() == sharded
    dp.bucket_elems = 1001                          # several buckets per matrix, with indivisible tails
    seen_sparse = []
    real_sparse = dp.exchange_sparse_rows

    def spy_sparse(st, name, *args, **kwargs):
        done = real_sparse(st, name, *args, **kwargs)
        seen_sparse.append((name, done))
        return done
    dp.exchange_sparse_rows = spy_sparse
    model = _model("cuda:{}".format(local))
    store = model.tf_manager.sessions[0].store
    if rank == 1:
        store.theta.add_(0.5)                       # replicas start from rank 0's variables
    dp.broadcast_parameters(store)
    shard = dp.shard(_batch())
    assert len(shard) == BATCH // world
    seen_early = []
    real_early = dp.all_reduce_early

    def spy(st, names):
        seen_early.append(list(names))
        return real_early(st, names)
    dp.all_reduce_early = spy
    losses, grad1 = [], None
    for step in range(3):
        res = model.tf_manager.execute(shard, model.trainer.feedables, [model.trainer], train=True)[0]
        losses.append(res.losses["decoder - cost"])
        if step == 0:
            grad1 = store.ensure_grad().cpu().numpy().copy()      # summed over ranks + L2 term, before clipping
    torch.cuda.synchronize()
    # two early spans per step: the vocabulary projection and the decoder embeddings (rows instead when sparse)
    assert len(seen_early) == (3 if sparse else 6) and "decoder/state_to_word_W" in seen_early[0]
    # both embedding matrices travel as rows: the decoder's first (its backward pass runs first), then the encoder's
    assert seen_sparse == ([("decoder/word_embeddings", True), ("encoder_input/embedding_matrix_0", True)] * 3
                           if sparse else [])
    # sharded optimizer: a rank's gradient buffer holds the sum (+ the L2 term) on the slices it owns only
    owned = np.ones(store.total, bool)
    if sharded:
        plan = dp.plan(store)
        owned[:] = False
        for lo, hi in plan.owned(rank) + plan.tails():
            owned[lo:hi] = True
        assert 0.3 * store.total < owned.sum() < 0.7 * store.total
        dp.gather_optimizer_slots(store, *store.ensure_adam())
    m, v = store.ensure_adam()
    report = dp.exchange_report()
    assert report["optimizer"] == ("sharded" if sharded else "replicated")
    np.savez(os.path.join(out_dir, "rank{}.npz".format(rank)), theta=store.theta.cpu().numpy(),
             losses=np.asarray(losses), grad1=grad1, owned=owned, m=m.cpu().numpy(), v=v.cpu().numpy())
    distributed.shutdown()


def _ranks_against_one_process(tmp_path, world, backend, sparse=False, sharded=True):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), backend, sparse, sharded), nprocs=world, join=True)
    ranks = [dict(np.load(tmp_path / "rank{}.npz".format(r))) for r in range(world)]
    r0 = ranks[0]
    for other in ranks[1:]:
        assert np.array_equal(r0["theta"], other["theta"]), "replicas diverged"
        assert np.array_equal(r0["m"], other["m"]) and np.array_equal(r0["v"], other["v"]), "gathered slots differ"
    # the summed gradient of step 1, from the slices their owners hold
    assert np.all(sum(r["owned"].astype(int) for r in ranks) >= 1)
    grad1 = np.zeros_like(r0["grad1"])
    for r in reversed(ranks):
        grad1[r["owned"]] = r["grad1"][r["owned"]]
    r0["grad1"] = grad1
    # the reference point: one process, whole batch, same seed
    model = _model()
    store = model.tf_manager.sessions[0].store
    start = store.theta.cpu().numpy().copy()
    full = _batch()
    model.tf_manager.execute(full, model.trainer.feedables, [model.trainer], train=True)
    want_g = store.ensure_grad().cpu().numpy().copy()
    scale = np.abs(want_g).max()
    assert scale > 0
    err = np.abs(r0["grad1"] - want_g).max()
    assert err <= 2e-4 * scale, (err, scale)          # fp32 re-association between shard sums and the full batch
    for _ in range(2):
        model.tf_manager.execute(full, model.trainer.feedables, [model.trainer], train=True)
    want = store.theta.cpu().numpy()
    # Adam normalises every element's step to ~lr, so elements whose gradient is rounding noise may step the
    # other way: the variables agree to within a few steps' worth, the gradients above are the sharp check
    assert np.abs(want - start).max() > 1e-4
    assert np.abs(r0["theta"] - want).max() <= 6.5e-4
    assert np.median(np.abs(r0["theta"] - want)) <= 1e-6
    assert r0["losses"].shape == (3,) and np.all(np.isfinite(r0["losses"])) and r0["losses"][2] < r0["losses"][0]
    return r0


def test_two_ranks_equal_one_process_on_the_full_batch(tmp_path):
    """... with the sharded optimizer (reduce-scatter -> update of the rank's slices -> all-gather: the default) and
    with the replicated one.  Within a run the replicas are bit-identical; the two runs agree to rounding only (the
    embedding gradients are scattered with float atomics, no two runs add them in the same order) -- that the sharded
    update IS the replicated one bit for bit is tests/test_distributed_cpu.py's, on deterministic gradients."""
    (tmp_path / "s").mkdir()
    (tmp_path / "r").mkdir()
    sharded = _ranks_against_one_process(tmp_path / "s", 2, "gloo", sharded=True)
    replicated = _ranks_against_one_process(tmp_path / "r", 2, "gloo", sharded=False)
    diff = np.abs(sharded["theta"] - replicated["theta"])
    assert diff.mean() < 2e-8 and diff.max() < 3e-5, (diff.mean(), diff.max())
    assert np.abs(sharded["m"] - replicated["m"]).max() <= 1e-4 * np.abs(replicated["m"]).max()


def test_two_ranks_with_the_encoder_embeddings_exchanged_as_rows(tmp_path):
    """NM_DP_SPARSE_EMB=1: the encoder's AND the decoder's embedding gradients travel as (unique ids, rows) --
    gathered, cancelled and re-added by libnmhip kernels in rank order -- instead of through the dense collective
    (under the sharded optimizer their buckets are then complete on every rank and skip the reduce-scatter); same
    identities:
    replicas bit-identical, summed gradient == one process on the full batch."""
    _ranks_against_one_process(tmp_path, 2, "gloo", sparse=True)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: one RCCL rank per GPU")
@pytest.mark.parametrize("world", [2, 4, 8])
def test_ranks_over_rccl_equal_one_process_on_the_full_batch(tmp_path, world):
    """The same identity with the collectives on RCCL over xGMI, one rank per GPU: replicas bit-identical after
    three optimizer steps, the summed gradient of step 1 equal to the full-batch gradient of one process, both
    early spans issued from inside the backward pass.  Runs wherever >= `world` GPUs are visible (skipped on the
    single-GPU boxes the round's tests run on)."""
    if torch.cuda.device_count() < world or BATCH % world:
        pytest.skip("{} GPUs visible".format(torch.cuda.device_count()))
    _ranks_against_one_process(tmp_path, world, "nccl", sparse=(world == 4))


def _rccl_worker(rank, world, port, out_dir, allreduce="torch"):
    replicated = allreduce.endswith("-replicated")
    allreduce = allreduce.split("-")[0]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      NM_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", NM_DIST_ALLREDUCE=allreduce,
                      NM_DP_SHARDED="0" if replicated else "1", NM_DP_BIG_VARIABLE="1500")
    os.environ.pop("NM_DIST_BACKEND", None)             # default on a GPU box: nccl = RCCL
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from neuralmonkey_amd import distributed
    dp = distributed.init_from_env()
    assert dp is not None and dp.world_size == 1 and dist.get_backend() == "nccl" and dp.forced
    assert (dp._comm is not None) == (allreduce == "nmhip")
    sharded = dp.sharded_active()
    assert sharded == (allreduce == "torch" and not replicated)      # (the library's communicator only all-reduces)
    dp.bucket_elems = 1001
    reduced = []
    real_span, real_bucket = dp._reduce_span, dp._reduce_bucket
    dp._reduce_span = lambda grad, lo, hi: (reduced.append((lo, hi)), real_span(grad, lo, hi))[1]
    dp._reduce_bucket = lambda grad, plan, idx: (reduced.append(plan.buckets[idx][:2]), real_bucket(grad, plan, idx))[1]
    model = _model()
    store = model.tf_manager.sessions[0].store
    dp.broadcast_parameters(store)
    batch = dp.shard(_batch())
    assert len(batch) == BATCH
    losses = []
    for _ in range(3):
        res = model.tf_manager.execute(batch, model.trainer.feedables, [model.trainer], train=True)[0]
        losses.append(res.losses["decoder - cost"])
    torch.cuda.synchronize()
    # every step: two early spans from inside the backward pass + the rest of the flat buffer, all of it covered
    assert len(reduced) >= 3 * 3
    assert sum(hi - lo for lo, hi in reduced) == 3 * store.ensure_grad().numel()
    np.savez(os.path.join(out_dir, "rccl.npz"), theta=store.theta.cpu().numpy(), losses=np.asarray(losses))
    distributed.shutdown()


@pytest.mark.parametrize("allreduce", ["torch", "torch-replicated", "nmhip"])
def test_rccl_process_group_of_one_trains_like_no_process_group(tmp_path, allreduce):
    """The RCCL path itself (backend nccl: process group, broadcast, early + bucketed collectives on their streams,
    global token count) with a world of one, where every collective is the identity: three optimizer steps end
    bit for bit where the same model ends without a process group.  ``torch``: the sharded optimizer (in-place
    reduce-scatter of every bucket, update, in-place all-gather: the default); ``torch-replicated``: all-reduce +
    the whole update (NM_DP_SHARDED=0); ``nmhip``: the gradient buckets go through the library's own communicator
    (nm_allreduce_*, NM_DIST_ALLREDUCE=nmhip) instead of torch.distributed's."""
    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path), allreduce), nprocs=1, join=True)
    got = np.load(tmp_path / "rccl.npz")
    model = _model()
    store = model.tf_manager.sessions[0].store
    full = _batch()
    want_losses = [model.tf_manager.execute(full, model.trainer.feedables, [model.trainer], train=True)[0]
                   .losses["decoder - cost"] for _ in range(3)]
    assert np.allclose(got["losses"], want_losses, rtol=1e-5, atol=0)
    diff = np.abs(got["theta"] - store.theta.cpu().numpy())       # (embedding-gradient atomics: see the test above)
    assert diff.max() <= 6.5e-4 and np.median(diff) <= 1e-6


def test_library_communicator_of_one_rank(dev):
    """nm_allreduce_* straight through the C ABI: id, communicator of one rank, two buckets enqueued behind the
    kernels that fill them, a device-side wait, destroy.  A sum over one rank is the identity; what is checked is the
    ordering against the caller's stream (the buckets are written right before and read right after)."""
    import ctypes
    from neuralmonkey_amd import _lib
    lib = _lib.load()
    uid = ctypes.create_string_buffer(128)
    _lib.check(lib.nm_allreduce_unique_id(uid, 128), "nm_allreduce_unique_id")
    assert any(uid.raw)
    comm = ctypes.c_void_p()
    _lib.check(lib.nm_allreduce_init(0, 1, uid, ctypes.byref(comm)), "nm_allreduce_init")
    stream = torch.cuda.current_stream().cuda_stream
    buf = torch.empty(3_000_000, device=dev)
    for rnd in range(3):
        buf.copy_(torch.arange(buf.numel(), device=dev, dtype=torch.float32) * (rnd + 1))
        _lib.check(lib.nm_allreduce_bucket(comm, stream, buf.data_ptr(), 1_000_000), "nm_allreduce_bucket")
        _lib.check(lib.nm_allreduce_bucket(comm, stream, buf.data_ptr() + 4_000_000, 2_000_000), "nm_allreduce_bucket")
        _lib.check(lib.nm_allreduce_wait(comm, stream), "nm_allreduce_wait")
        total = float(buf.double().sum())
        n = buf.numel()
        assert total == (rnd + 1) * n * (n - 1) / 2
    assert lib.nm_allreduce_bucket(comm, stream, None, 10) != 0           # refused, with a message
    assert b"empty bucket" in lib.nm_last_error()
    _lib.check(lib.nm_allreduce_destroy(comm), "nm_allreduce_destroy")
