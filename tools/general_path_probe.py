"""bench.py's ``configs.general_path`` leg alone, once per setting of a switch:
    python tools/general_path_probe.py NM_NEMATUS_CLUSTER 1 0
(prints one JSON object per value; used to price the one-launch NematusGRU loops against the step-by-step tape)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    train_only = "--train-only" in sys.argv
    argv = [a for a in sys.argv[1:] if a != "--train-only"]
    name, values = (argv[0], argv[1:]) if len(argv) > 1 else ("NM_NEMATUS_CLUSTER", ["1"])
    sys.argv = sys.argv[:1]
    import bench
    args = bench.parse()
    args.general_train_only = train_only
    for v in values:
        os.environ[name] = v
        leg = bench.general_path_leg(args, "cuda:0")
        leg.pop("workload", None)
        print(name + "=" + v, json.dumps({k: (round(x, 3) if isinstance(x, float) else x) for k, x in leg.items()}))


if __name__ == "__main__":
    main()
