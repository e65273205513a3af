#!/bin/bash
# the ingest code under AddressSanitizer + UBSan on the CPU (GPU sanitizers are not available on the pool):  bash tools/sanitize/run.sh
set -eu
cd "$(dirname "$0")/../.."
out=${TMPDIR:-/tmp}/crass_sanitize; rm -rf $out; mkdir -p $out
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude crass_amd/csrc/ingest.cpp crass_amd/csrc/pgzip.cpp tools/sanitize/ingest_main.cpp -o $out/ingest_asan -lz -ldl -lpthread
python3 - $out <<'P'
import gzip, os, random, struct, sys, zlib
out = sys.argv[1]
rng = random.Random(99)
def soup(fq, nasty, n):
    parts = []
    eol = "\r\n" if rng.random() < 0.15 else "\n"
    wrap = rng.choice([0, 0, 7, 60])
    for i in range(n):
        L = rng.choice([0, 1, 5, 16, 17, 31, 64, rng.randint(1, 200)])
        alpha = rng.choice(["ACGT", "ACGTN", "acgtACGT", "ACGT>", "ACGT+", "ACG T"] if nasty else ["ACGT", "ACGT", "ACGTN"])
        s = "".join(rng.choice(alpha) for _ in range(L))
        body = eol.join(s[k:k + wrap] for k in range(0, len(s), wrap)) if wrap and s else s
        if fq:
            q = "".join(rng.choice("IIIH5#!~@+>") for _ in range(max(0, len(s.replace(" ", "")) + (rng.choice([0, 0, -1, 2]) if nasty else 0))))
            parts.append("@r%d c\n%s%s+%s%s%s" % (i, body, eol, eol, q, eol))
        else:
            parts.append(">r%d%s%s%s" % (i % 17, eol, body, eol))
    t = "".join(parts)
    return t[:max(1, len(t) - rng.randint(0, 20))] if nasty else t
k = 0
for fq in (False, True):
    for nasty in (False, True):
        for n in (1, 30, 3000):
            for rep in range(3):
                t = soup(fq, nasty, n).encode()
                open(os.path.join(out, "s%03d.txt" % k), "wb").write(t)
                open(os.path.join(out, "s%03d.gz" % k), "wb").write(gzip.compress(t, rng.choice([1, 6, 9])))
                k += 1
# a few megabytes for the several-thread inflate and BGZF
big = "".join(">b%d\n%s\n" % (i, "".join(rng.choice("ACGT") for _ in range(150))) for i in range(40000)).encode()
open(os.path.join(out, "big.gz"), "wb").write(gzip.compress(big, 6))
with open(os.path.join(out, "big.bgzf.gz"), "wb") as f:
    for i in list(range(0, len(big), 60000)) + [None]:
        c = b"" if i is None else big[i:i + 60000]
        co = zlib.compressobj(6, zlib.DEFLATED, -15); d = co.compress(c) + co.flush()
        f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", 12 + 6 + len(d) + 8 - 1) + d + struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c)))
raw = bytearray(gzip.compress(big, 6)); raw[len(raw) // 2] ^= 0x20
open(os.path.join(out, "damaged.gz"), "wb").write(bytes(raw))
P
export ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0
export CRASS_FASTX_CHUNK=200 CRASS_PGZIP_MIN_BYTES=1000 CRASS_PGZIP_CHUNK_BYTES=30000
$out/ingest_asan $out/*.txt $out/*.gz > $out/report.txt 2> $out/sanitizer.txt || true
grep -c "^ok  " $out/report.txt; grep "^DIFF" $out/report.txt | head; grep -c "ERROR: AddressSanitizer\|runtime error\|LeakSanitizer" $out/sanitizer.txt || true; head -40 $out/sanitizer.txt
# ... and under ThreadSanitizer (the stream's read-ahead thread, the sharded name tables, the several-thread inflate, the index's threads)
g++ -std=c++17 -O1 -g -fsanitize=thread -fno-omit-frame-pointer -Iinclude crass_amd/csrc/ingest.cpp crass_amd/csrc/pgzip.cpp tools/sanitize/ingest_main.cpp -o $out/ingest_tsan -lz -ldl -lpthread
$out/ingest_tsan $out/s00*.txt $out/s01*.gz $out/s03*.txt $out/big.gz $out/big.bgzf.gz > $out/report_tsan.txt 2> $out/tsan.txt || true
echo "tsan: $(grep -c '^ok  ' $out/report_tsan.txt) files ok, $(grep -c 'WARNING: ThreadSanitizer' $out/tsan.txt || true) warnings"
