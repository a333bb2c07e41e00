"""Generates tests/golden/bpe_golden.json by running the REFERENCE's own BPE code
(/root/reference/lib/subword_nmt/apply_bpe.py via neuralmonkey/processors/bpe.py semantics) on its
fixture merge table tests/data/merges_100.bpe and sentences of tests/data/train.tc.en.
Run in the build container only (the reference tree does not travel to the GPU box):

    python tests/golden/make_bpe_golden.py
"""
import importlib.util
import json
import os

REF = "/root/reference"
spec = importlib.util.spec_from_file_location("ref_apply_bpe", os.path.join(REF, "lib/subword_nmt/apply_bpe.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

with open(os.path.join(REF, "tests/data/merges_100.bpe"), encoding="utf-8") as handle:
    merges = [line.rstrip("\n") for line in handle]
bpe = mod.BPE(merges, "@@")
with open(os.path.join(REF, "tests/data/train.tc.en"), encoding="utf-8") as handle:
    sentences = [line.split() for _, line in zip(range(40), handle)]
sentences.append(["", "a", "unsegmentable-ζ", "the"])        # the empty-token pass-through of bpe.py:33-36


def reference_preprocess(sentence):
    out = []
    for word in sentence:
        if not word:
            out.append(word)
            continue
        pieces = mod.encode(word, bpe.bpe_codes)
        out.extend(p + bpe.separator for p in pieces[:-1])
        out.append(pieces[-1])
    return out


golden = {"merges": merges, "separator": "@@",
          "cases": [{"input": s, "output": reference_preprocess(s)} for s in sentences]}
path = os.path.join(os.path.dirname(__file__), "bpe_golden.json")
with open(path, "w", encoding="utf-8") as handle:
    json.dump(golden, handle, ensure_ascii=False, indent=0)
print("wrote", path, len(golden["cases"]), "cases")
