// tests/hostemu/parser_check.cpp — TEST INFRASTRUCTURE: the product's sequential reader (mcx_files.cpp's Parser: .gz through the parallel inflater, BGZF,
// zlib's reader, FASTA with multi-line records) on the host, without a GPU: every record it hands out, as name <TAB> bases <TAB> qualities lines.
// The file is compiled WITH the product's source (the class lives in an anonymous namespace there) and linked against libmcx.so for the entry points that
// source refers to; nothing of the mapping path runs.
#include "../../mapcaller_amd/csrc/mcx_files.cpp"

extern "C" long long parser_dump(const char *path, int max_len, int per_take, const char *out_path, char *err, int err_cap)
{
    Parser ps;
    std::string e;
    if (!ps.open(path, e)) { snprintf(err, (size_t)err_cap, "%s", e.c_str()); return -1; }
    FILE *f = fopen(out_path, "wb");
    if (!f) return -2;
    long long n = 0;
    for (bool more = true; more;) {
        View v;
        more = ps.take(v, (uint32_t)per_take, max_len);
        if (!v.error.empty()) { snprintf(err, (size_t)err_cap, "%s", v.error.c_str()); fclose(f); return -3; }
        for (const Rec &r : v.recs) {
            fwrite(v.base + r.name, 1, r.name_len, f); fputc('\t', f);
            fwrite(v.base + r.seq, 1, r.rlen, f); fputc('\t', f);
            if (ps.fastq()) fwrite(v.base + r.qual, 1, r.q_take, f);
            fputc('\n', f);
            n++;
        }
    }
    fclose(f);
    return n;
}
