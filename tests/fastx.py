"""Pure-Python FASTA/FASTQ(.gz) record reader with klib kseq semantics
(reference: src/crass/kseq.cpp:171-226 as driven by libcrispr.cpp:96-131).
Test helper; the product's own reader is crass_amd's C++ one and is checked against this
and against the compiled reference kseq (oracle/_ref)."""
import gzip


def _isspace(c):
    return c in b" \t\n\v\f\r"


def read_fastx(path):
    """Returns list of (name, comment_or_None, seq, qual_or_None) as bytes.
    comment/qual follow the *stale pointer* semantics of searchFile: once any record had a
    comment (qual), later records without one see the stale buffer contents; we reproduce
    kseq's buffers: comment.l is reset but comment.s keeps old bytes ... which searchFile then
    reads as a C string — i.e. the previous comment's text (kseq.cpp:182-186)."""
    with gzip.open(path, "rb") as f:
        data = f.read()
    n = len(data)
    pos = 0
    recs = []
    last_char = 0
    comment_buf = None   # stale kstring contents (bytes) or None if never allocated
    qual_buf = None
    while True:
        if last_char == 0:
            while pos < n and data[pos] not in b">@":
                pos += 1
            if pos >= n:
                break
            last_char = data[pos]
            pos += 1
        # name: until whitespace
        if pos >= n:
            break
        st = pos
        while pos < n and not _isspace(data[pos:pos + 1]):
            pos += 1
        name = data[st:pos]
        c = data[pos] if pos < n else None
        pos += 1
        if c is not None and c != 0x0A:
            st = pos
            while pos < n and data[pos] != 0x0A:
                pos += 1
            comment_buf = data[st:pos]
            pos += 1
        # sequence
        seq = bytearray()
        c = None
        while pos < n:
            c = data[pos]
            pos += 1
            if c in b">+@":
                break
            if 33 <= c <= 126:
                seq.append(c)
            c = None
        if c in (0x3E, 0x40):
            last_char = c
        qual = None
        if c == 0x2B:  # '+'
            while pos < n and data[pos] != 0x0A:
                pos += 1
            if pos >= n:
                break        # truncated: kseq returns -2, loop ends
            pos += 1
            q = bytearray()
            while pos < n and len(q) < len(seq):
                ch = data[pos]
                pos += 1
                if 33 <= ch <= 127:
                    q.append(ch)
            # kseq consumes one more char in its while condition when it stops on length
            if pos < n and len(q) >= len(seq):
                pos += 1
            last_char = 0
            if len(q) != len(seq):
                break
            qual_buf = bytes(q)
        elif qual_buf is not None:
            # stale: qual.l reset to 0 and never re-terminated before return at kseq.cpp:201-202?
            # seq->qual.l = 0 but buffer untouched => C string = old contents
            pass
        recs.append((bytes(name), comment_buf, bytes(seq), qual_buf))
        if c is None and pos >= n:
            break
    return recs
