"""Writes ERR091571.qname_flag_seq.tsv.gz from the reference's golden/ERR091571.sam (the SAM text of ERR091571.bam, both
data files of the reference's tests): one line per alignment record, "qname<TAB>flag<TAB>seq".  Run in the build
container only (reads /root/reference)."""
import gzip
import os

here = os.path.dirname(os.path.abspath(__file__))
with open("/root/reference/golden/ERR091571.sam") as f, gzip.open(os.path.join(here, "ERR091571.qname_flag_seq.tsv.gz"), "wt") as out:
    for line in f:
        if line.startswith("@"):
            continue
        c = line.rstrip("\n").split("\t")
        out.write("%s\t%s\t%s\n" % (c[0], c[1], c[9]))
