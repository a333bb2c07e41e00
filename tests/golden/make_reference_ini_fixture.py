"""Bundle the reference's OWN acceptance configs and the data they name into one fixture.

    python tests/golden/make_reference_ini_fixture.py        (in the build container)

/root/reference does not exist on the GPU box, so the byte-for-byte INI files of SURVEY 4.1's drop-in
acceptance list (tests/{small,beamsearch,bahdanau,transformer,flat-multiattention,factored,
beamsearch_ensembles}.ini, the neuralmonkey-run data configs of tests/tests_run.sh) and the small data
files they reference (tests/data: parallel text, vocabularies, the 13 pre-extracted 8x8x2048 ResNet maps
of tests/data/flickr30k) travel as tests/golden/reference_tests.tar.gz.  Nothing is edited: the
archive members are the reference's bytes (tests/test_reference_inis.py compares them with
/root/reference when that exists).  Configs and data only -- no reference source code.
"""
import glob
import io
import os
import tarfile

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "reference_tests.tar.gz")

INIS = ["small", "beamsearch", "bahdanau", "transformer", "flat-multiattention", "factored",
        "beamsearch_ensembles", "test_data", "test_data_ensembles_single", "test_data_ensembles_duplicate",
        "test_data_ensembles_all"]
DATA = ["train.tc.en", "train.tc.de", "val.tc.en", "val.tc.de", "val10.part1.tc.en", "val10.part2.tc.en",
        "val10.tc.en", "val10.tc.de", "encoder_vocab.tsv", "decoder_vocab.tsv", "factored_surface_vocab.tsv",
        "factored_tag_vocab.tsv", "factored_decoder_vocab.tsv", "multi/*.txt", "flickr30k/*.en",
        "flickr30k/*.de", "flickr30k/*.txt", "flickr30k/*.npz"]


# A second, small archive (added later; the first one is 6 MB of feature maps and stays as it is): the post-editing
# configuration of tests/tests_run.sh:12 with its data, and tests/nematus.ini, which the reference itself refuses.
OUT_MORE = os.path.join(HERE, "reference_tests_more.tar.gz")
INIS_MORE = ["post-edit", "nematus"]
DATA_MORE = ["postedit/*", "postedit_target_vocab.tsv"]


def members(inis=INIS, data=DATA):
    for name in inis:
        yield "tests/{}.ini".format(name)
    for pattern in data:
        for path in sorted(glob.glob(os.path.join(REF, "tests/data", pattern))):
            yield os.path.relpath(path, REF)


def write(out, rels):
    with tarfile.open(out, "w:gz", compresslevel=9) as tar:
        for rel in rels:
            with open(os.path.join(REF, rel), "rb") as fh:
                data = fh.read()
            info = tarfile.TarInfo(rel)
            info.size = len(data)
            info.mtime = 0
            tar.addfile(info, io.BytesIO(data))
    print(out, os.path.getsize(out))


def main(which):
    if "first" in which:
        write(OUT, members())
    if "more" in which:
        write(OUT_MORE, members(INIS_MORE, DATA_MORE))


if __name__ == "__main__":
    import sys
    main(sys.argv[1:] or ["first", "more"])
