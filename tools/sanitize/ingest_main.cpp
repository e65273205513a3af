// AddressSanitizer / UBSan harness for the host-only ingest code (csrc/ingest.cpp + csrc/pgzip.cpp; GPU sanitizers are not available on
// the pool): every input file given is read by the whole-file reader, the streamed reader (tiny chunks) and the index (built, all
// records fetched from the mapping, the mapping dropped, fetched again from the file); gzip'd inputs go through the BGZF / several-thread /
// serial inflate as their format and the environment decide.  Prints one line per file; any sanitizer report fails the run.
#include "../../include/crass_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
int main(int argc, char **argv)
{
    int bad = 0;
    for (int a = 1; a < argc; a++) {
        crass_fastx fx;
        const int r1 = crass_read_fastx(argv[a], &fx);
        unsigned long long n1 = r1 == CRASS_OK ? fx.n_reads : 0, seq1 = r1 == CRASS_OK && fx.n_reads ? fx.seq_off[fx.n_reads] : 0;
        if (r1 == CRASS_OK) crass_free_fastx(&fx);
        unsigned long long n2 = 0, seq2 = 0;
        crass_name_table *nt = crass_name_table_create();
        crass_fastx_stream *st = nullptr;
        int r2 = crass_fastx_stream_open(argv[a], 300, nt, 0, &st);
        if (r2 == CRASS_OK) {
            for (;;) { crass_fastx c; r2 = crass_fastx_stream_next(st, &c); if (r2 != CRASS_OK || c.n_reads == 0) break; n2 += c.n_reads; seq2 += c.seq_off[c.n_reads]; }
            crass_fastx_stream_close(st);
        }
        crass_name_table_destroy(nt);
        crass_fastx_index *ix = nullptr;
        const int r3 = crass_index_fastx(argv[a], &ix);
        unsigned long long n3 = 0, seq3 = 0, seq4 = 0;
        if (r3 == CRASS_OK) {
            crass_reads rd; uint32_t ml = 0; int lr = 0;
            crass_fastx_index_reads(ix, &rd, &ml, &lr);
            n3 = rd.n_reads;
            std::vector<uint64_t> all(n3);
            for (uint64_t i = 0; i < n3; i++) all[i] = n3 - 1 - i;
            crass_fastx f;
            if (crass_fastx_index_fetch(ix, all.data(), n3, &f) == CRASS_OK) { seq3 = n3 ? f.seq_off[n3] : 0; crass_free_fastx(&f); }
            crass_fastx_index_drop_text(ix);
            if (crass_fastx_index_fetch(ix, all.data(), n3, &f) == CRASS_OK) { seq4 = n3 ? f.seq_off[n3] : 0; crass_free_fastx(&f); }
            crass_fastx_index_free(ix);
        }
        const bool ok = (r1 != CRASS_OK || (n1 == n2 && seq1 == seq2)) && (r3 != CRASS_OK || (n3 == n1 && seq3 == seq1 && seq4 == seq1));
        printf("%s %s: whole rc %d, %llu records %llu bases; stream rc %d, %llu / %llu; index rc %d, %llu / %llu / %llu\n", ok ? "ok  " : "DIFF", argv[a], r1, n1, seq1, r2, n2, seq2, r3, n3, seq3, seq4);
        bad += ok ? 0 : 1;
    }
    return bad ? 1 : 0;
}
