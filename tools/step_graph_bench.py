import sys, time, torch
sys.path.insert(0, "/root/repo")
from tests import test_general_gpu as TG, test_transformer_gpu as TT
dev = torch.device("cuda:0")
def run(build, data, graphs):
    m = build()
    sess = m["tfm"].sessions[0]; sess.use_step_graphs = graphs
    ds = data(64, 20, 20, 24, seed=1)[0]
    for _ in range(3): m["tfm"].execute(ds, m["trainer"].feedables, [m["trainer"]], train=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(30): m["tfm"].execute(ds, m["trainer"].feedables, [m["trainer"]], train=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / 30 * 1e3
cfg, es, et = TG.CASES["small_ini"]
for g in (False, True):
    print("small_ini graphs", g, "%.2f ms/step" % run(lambda: TG._build(dev, cfg, es, et, max_len=24), TG._data, g))
c2, d, ff = TT.CASES["wide"]
for g in (False, True):
    print("transformer wide graphs", g, "%.2f ms/step" % run(lambda: TT._build(dev, c2, d, ff, max_len=24), TT._data, g))
