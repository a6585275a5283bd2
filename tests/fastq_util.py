"""Minimal FASTQ reader for the fixtures (test helper; record semantics of SURVEY.md A.6)."""


def read_fastq(path):
    """-> list of (header_line, id, seq, quals) with trailing whitespace stripped."""
    recs = []
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    i = 0
    while i + 3 < len(lines) + 1 and i < len(lines):
        h = lines[i].rstrip()
        if not h:
            break
        seq = lines[i + 1].rstrip()
        quals = lines[i + 3].rstrip()
        rid = h[1:].split(b" ")[0].split(b"\t")[0]
        recs.append((h, rid, seq, quals))
        i += 4
    return recs


def write_fastq(path, records, qual=b"I"):
    """records: iterable of (id str, sequence bytes); constant qualities."""
    with open(path, "wb") as f:
        for rid, seq in records:
            f.write(b"@" + rid.encode() + b"\n" + seq + b"\n+\n" + qual * len(seq) + b"\n")
