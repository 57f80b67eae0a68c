// oracle/ref_shim.cpp — TEST INFRASTRUCTURE ONLY.
//
// Line-oriented driver around the *real* reference objects (compiled by oracle/Makefile from
// /root/reference/src where they lie; this file is ours), built as oracle/_ref/mcref_tool.  It
// lets tests and the golden-vector generator call the reference's own functions one at a time
// (an executable rather than a shared library: the BAM-only htslib symbols that
// ReadMapping.cpp references stay unresolved, which a dlopen would refuse):
//
//   BWT_Search       (reference src/bwt_search.cpp:121, declared src/structure.h:279)
//   nw_alignment     (src/nw_alignment.cpp:18,   declared src/structure.h:289)
//   ksw2_alignment   (src/ksw2_alignment.cpp:250, declared src/structure.h:292)
//   ksw_extz2_sse    (src/ksw2_alignment.cpp:70; non-static, gives ez.score)
//   bwa_idx_load / RestoreReferenceInfo (src/bwt_index.cpp:150,232)
//
// Nothing here restates the algorithm; the restatement is oracle/mcx_oracle.cpp.
#include "structure.h"

// ksw2_alignment.cpp keeps this struct private (ksw2_alignment.cpp:11-17); same layout here.
// The typedef name is what the symbol is mangled with, so it must be spelled identically.
typedef struct {
    uint32_t max;
    int max_q, max_t;
    int mqe, mqe_t;
    int mte, mte_q;
    int score;
} ksw_extz_t;
extern string ksw_extz2_sse(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                            int8_t m, int8_t q, int8_t e, int w, ksw_extz_t *ez);
extern bwtint_t bwt_sa(bwtint_t k);
extern float MaxMisMatchRate;

static bool g_loaded = false;

extern "C" {

// Load an index exactly as main() does (main.cpp:350-361) and set the defaults of main.cpp:159-191.
int mcref_load_index(const char *prefix)
{
    if (g_loaded) return 0;
    iThreadNum = 1; MaxPosDiff = 30; MaxMisMatchRate = 0.05f; NW_ALG = true; bUnique = true;
    RefIdx = bwa_idx_load(prefix);
    if (RefIdx == 0) return -1;
    Refbwt = RefIdx->bwt;
    RestoreReferenceInfo();
    g_loaded = true;
    return 0;
}

long long mcref_genome_size() { return (long long)GenomeSize; }

// seq: codes 0..4.  loc must hold 50 entries.
int mcref_bwt_search(const uint8_t *seq, int start, int stop, int *len, int *freq, uint64_t *loc)
{
    bwtSearchResult_t r = BWT_Search((uint8_t *)seq, start, stop);
    *len = r.len; *freq = r.freq;
    for (int i = 0; i < r.freq; i++) loc[i] = r.LocArr[i];
    if (r.freq > 0) delete[] r.LocArr;
    return 0;
}

unsigned long long mcref_bwt_sa(unsigned long long k) { return bwt_sa(k); }

static int run_aln(bool nw, const char *s1, int m, const char *s2, int n, char *o1, char *o2, int cap)
{
    string a(s1, m), b(s2, n);
    if (nw) nw_alignment(m, a, n, b); else ksw2_alignment(m, a, n, b);
    if ((int)a.length() >= cap || (int)b.length() >= cap) return -1;
    memcpy(o1, a.c_str(), a.length() + 1);
    memcpy(o2, b.c_str(), b.length() + 1);
    return (int)a.length() == (int)b.length() ? (int)a.length() : -2;
}

int mcref_nw(const char *s1, int m, const char *s2, int n, char *o1, char *o2, int cap)
{ return run_aln(true, s1, m, s2, n, o1, o2, cap); }

int mcref_ksw2(const char *s1, int m, const char *s2, int n, char *o1, char *o2, int cap)
{ return run_aln(false, s1, m, s2, n, o1, o2, cap); }

// q/t: codes 0..4 (query = read fragment, target = genome fragment), parameters exactly as
// ksw2_alignment passes them (ksw2_alignment.cpp:260).  ops receives the *reversed* op string
// (M/I/D) that ksw_backtrack returns; returns its length, *score = ez.score.
int mcref_ksw2_extz(const uint8_t *q, int qlen, const uint8_t *t, int tlen, int *score, char *ops, int cap)
{
    ksw_extz_t ez;
    string c = ksw_extz2_sse(qlen, q, tlen, t, 5, 2, 1, -1, &ez);
    *score = ez.score;
    if ((int)c.length() >= cap) return -1;
    memcpy(ops, c.c_str(), c.length() + 1);
    return (int)c.length();
}

}

// Runs the reference's own Mapping() (ReadMapping.cpp:689) with -vcf bookkeeping on and dumps what
// UpdateProfile / UpdateMultiHitCount (AlignmentProfile.cpp:41-271) accumulated:
//   <out>.prof : GenomeSize records of 10 x u16 {A,C,G,T,multi_hit,readCount,F1,R2,F2,R1}
//   <out>.maps : text lines "I pos seq n" / "D pos seq n" (InsertSeqMap/DeleteSeqMap), "B pos n"
//                (BreakPointMap), "V gPos dist" / "T gPos dist" (InversionSiteVec/TranslocationSiteVec)
extern map<int64_t, uint16_t> BreakPointMap;
// sparse != 0 (request Q): the profile of a genome too large for the dense file — only the positions
//   with a non-zero counter, as records {int64 pos, 10 x u16} in <out>.prof.nz
static int dump_profile(const char *out, int sparse)
{
    string p = string(out) + (sparse ? ".prof.nz" : ".prof");
    FILE *f = fopen(p.c_str(), "wb");
    if (!f) return -1;
    for (int64_t g = 0; g < GenomeSize; g++) {
        const MappingRecord_t &m = MappingRecordArr[g];
        uint16_t v[10] = {(uint16_t)m.A, (uint16_t)m.C, (uint16_t)m.G, (uint16_t)m.T, (uint16_t)m.multi_hit, (uint16_t)m.readCount, m.F1, m.R2, m.F2, m.R1};
        if (sparse) {
            bool any = false;
            for (int k = 0; k < 10; k++) any = any || v[k] != 0;
            if (!any) continue;
            fwrite(&g, 8, 1, f);
        }
        fwrite(v, 2, 10, f);
    }
    fclose(f);
    p = string(out) + ".maps";
    f = fopen(p.c_str(), "w");
    for (map<int64_t, map<string, uint16_t> >::iterator a = InsertSeqMap.begin(); a != InsertSeqMap.end(); a++)
        for (map<string, uint16_t>::iterator b = a->second.begin(); b != a->second.end(); b++) fprintf(f, "I %lld %s %d\n", (long long)a->first, b->first.c_str(), (int)b->second);
    for (map<int64_t, map<string, uint16_t> >::iterator a = DeleteSeqMap.begin(); a != DeleteSeqMap.end(); a++)
        for (map<string, uint16_t>::iterator b = a->second.begin(); b != a->second.end(); b++) fprintf(f, "D %lld %s %d\n", (long long)a->first, b->first.c_str(), (int)b->second);
    for (map<int64_t, uint16_t>::iterator a = BreakPointMap.begin(); a != BreakPointMap.end(); a++) fprintf(f, "B %lld %d\n", (long long)a->first, (int)a->second);
    for (size_t i = 0; i < InversionSiteVec.size(); i++) fprintf(f, "V %lld %lld\n", (long long)InversionSiteVec[i].gPos, (long long)InversionSiteVec[i].dist);
    for (size_t i = 0; i < TranslocationSiteVec.size(); i++) fprintf(f, "T %lld %lld\n", (long long)TranslocationSiteVec[i].gPos, (long long)TranslocationSiteVec[i].dist);
    fclose(f);
    return 0;
}

static int run_profile(const char *fq1, const char *fq2, int ksw2, const char *out, int sparse = 0)
{
    ReadFileNameVec1.clear(); ReadFileNameVec2.clear();
    ReadFileNameVec1.push_back(fq1);
    if (fq2 && fq2[0]) ReadFileNameVec2.push_back(fq2);
    NW_ALG = !ksw2; bVCFoutput = true; bSAMoutput = false; iThreadNum = 1; iMaxDuplicate = 5; MaxClipSize = 5;
    static char logname[] = "/dev/null";
    LogFileName = logname;
    if (MappingRecordArr) delete[] MappingRecordArr;
    MappingRecordArr = new MappingRecord_t[GenomeSize]();
    InsertSeqMap.clear(); DeleteSeqMap.clear(); BreakPointMap.clear(); InversionSiteVec.clear(); TranslocationSiteVec.clear();
    pthread_mutex_init(&VarLock, NULL); pthread_mutex_init(&OutputLock, NULL); pthread_mutex_init(&LibraryLock, NULL); pthread_mutex_init(&ProfileLock, NULL);
    StartProcessTime = time(NULL);
    Mapping();
    return dump_profile(out, sparse);
}

// Request R: ONE run of the reference's own main() (main.cpp:154-395, compiled into this tool as mapcaller_ref_main) — index load, Mapping() with `-sam`,
// VariantCalling() with `-vcf` — with the accumulated profile and maps dumped in between: main_lib.o is compiled with -DVariantCalling=mcref_vc_hook
// (oracle/Makefile), so the call at main.cpp:380 arrives here first.  Saves the tests a second index load and a second mapping pass at 3.1 Gbp.
extern int mapcaller_ref_main(int argc, char *argv[]);
static string g_hook_out;
static int g_hook_sparse = 0, g_hook_rc = 0;
void mcref_vc_hook()
{
    if (!g_hook_out.empty()) g_hook_rc = dump_profile(g_hook_out.c_str(), g_hook_sparse);
    VariantCalling();
}

// stdin protocol, one request per line, one reply line each:
//   P <nw|ksw2> <out prefix> <fq1> [fq2]  -> "ok" after <out>.prof / <out>.maps are written
//   Q <nw|ksw2> <out prefix> <fq1> [fq2]  -> the same with <out>.prof.nz (non-zero positions only) in place of <out>.prof
//   R <nw|ksw2> <out prefix> <index prefix> <sam> <vcf> <fq1> [fq2] -> the reference's whole main() at -t 1 with -sam / -vcf, <out>.prof.nz / <out>.maps dumped
//                                 between Mapping() and VariantCalling() (a process of its own: main() loads and frees the index itself)
//   L <prefix>                 -> "ok <genome size>"
//   S <start> <codes 0-4>      -> "<len> <freq> <loc>..."            BWT_Search(seq, start, strlen)
//   D <q ascii> <t ascii>      -> "<nw a1> <nw a2> <ksw2 a1> <ksw2 a2> <ez.score> <ops reversed>"
int main()
{
    char *line = NULL;
    size_t cap = 0;
    ssize_t n;
    while ((n = getline(&line, &cap, stdin)) > 0) {
        while (n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
        if (line[0] == 'L') {
            int rc = mcref_load_index(line + 2);
            printf("%s %lld\n", rc == 0 ? "ok" : "fail", mcref_genome_size());
        } else if (line[0] == 'S') {
            int start = 0, used = 0;
            sscanf(line + 2, "%d %n", &start, &used);
            const char *digits = line + 2 + used;
            int L = (int)strlen(digits);
            std::vector<uint8_t> seq(L);
            for (int i = 0; i < L; i++) seq[i] = (uint8_t)(digits[i] - '0');
            int len, freq;
            uint64_t loc[64];
            mcref_bwt_search(seq.data(), start, L, &len, &freq, loc);
            printf("%d %d", len, freq);
            for (int i = 0; i < freq; i++) printf(" %llu", (unsigned long long)loc[i]);
            printf("\n");
        } else if (line[0] == 'D') {
            char *q = line + 2, *t = strchr(q, ' ');
            if (!t) { printf("bad\n"); fflush(stdout); continue; }
            *t++ = 0;
            int m = (int)strlen(q), k = (int)strlen(t), c = m + k + 8;
            std::vector<char> a1(c), a2(c), b1(c), b2(c), ops(c);
            mcref_nw(q, m, t, k, a1.data(), a2.data(), c);
            mcref_ksw2(q, m, t, k, b1.data(), b2.data(), c);
            std::vector<uint8_t> qc(m), tc(k);
            for (int i = 0; i < m; i++) qc[i] = nst_nt4_table[(uint8_t)q[i]];
            for (int i = 0; i < k; i++) tc[i] = nst_nt4_table[(uint8_t)t[i]];
            int score = 0;
            mcref_ksw2_extz(qc.data(), m, tc.data(), k, &score, ops.data(), c);
            printf("%s %s %s %s %d %s\n", a1.data(), a2.data(), b1.data(), b2.data(), score, ops.data());
        } else if (line[0] == 'P' || line[0] == 'Q') {
            char alg[16], out[1024], f1[1024], f2[1024];
            f2[0] = 0;
            int k = sscanf(line + 2, "%15s %1023s %1023s %1023s", alg, out, f1, f2);
            int rc = k >= 3 ? run_profile(f1, f2, strcmp(alg, "ksw2") == 0, out, line[0] == 'Q') : -1;
            printf("%s\n", rc == 0 ? "ok" : "fail");
        } else if (line[0] == 'R') {
            char alg[16], out[1024], idx[1024], sam[1024], vcf[1024], f1[1024], f2[1024];
            f2[0] = 0;
            int k = sscanf(line + 2, "%15s %1023s %1023s %1023s %1023s %1023s %1023s", alg, out, idx, sam, vcf, f1, f2);
            if (k < 6 || g_loaded) { printf("fail\n"); fflush(stdout); continue; } // (main() loads the index itself: not after L)
            g_hook_out = out; g_hook_sparse = 1; g_hook_rc = 0;
            static char a0[] = "MapCaller", ai[] = "-i", af[] = "-f", af2[] = "-f2", aa[] = "-alg", as[] = "-sam", av[] = "-vcf", at[] = "-t", one[] = "1", al[] = "-log", devnull[] = "/dev/null";
            std::vector<char *> argv = {a0, ai, idx, af, f1};
            if (f2[0]) { argv.push_back(af2); argv.push_back(f2); }
            for (char *x : {aa, alg, as, sam, av, vcf, at, one, al, devnull}) argv.push_back(x);
            int rc = mapcaller_ref_main((int)argv.size(), argv.data());
            g_hook_out.clear();
            printf("%s\n", rc == 0 && g_hook_rc == 0 ? "ok" : "fail");
        } else printf("bad\n");
        fflush(stdout);
    }
    return 0;
}
