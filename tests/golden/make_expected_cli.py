"""Generates tests/golden/expected_find_mems_*.txt: the find_mems stdout the reference would print
(timing trailer removed) for the reference's own fixture pair, computed by the CPU oracle.
RESTATEMENT-DERIVED (the reference binary cannot be built here; see DESIGN.md "Oracle")."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_ffi as O  # noqa: E402
from cli_format import format_find_mems  # noqa: E402

BT = os.path.join(HERE, "bidirectional_test")
ri = O.RIndex(os.path.join(BT, "xy.ri"))
tags = O.Tags(os.path.join(BT, "xy_bidirectional_compressed.tags"), O.TAGS_BYTECODE)
for reads_file, ml, mo in (("reads.txt", 5, 1), ("reads.txt", 3, 1), ("test_reads.txt", 3, 1)):
    reads = [l for l in open(os.path.join(BT, reads_file)).read().split("\n") if l]
    cat, offs = O.pack_reads(reads)
    res = O.find_mems_batch(ri, tags, cat, offs, ml, mo)
    name = "expected_find_mems_xy_%s_%d_%d.txt" % (reads_file.split(".")[0], ml, mo)
    open(os.path.join(HERE, name), "w").write(format_find_mems(res))
    print(name, len(res["mems"]), "MEMs")
