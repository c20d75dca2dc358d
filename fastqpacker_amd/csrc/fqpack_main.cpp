// fqpack — command-line driver over libfqzhip with the reference CLI's surface
// (cmd/fqpack/main.go:65-203): flags -d -i -o -c -b -w -version -h, two positionals,
// "-"/empty = stdin/stdout, gzip input detected by ".gz" suffix or 1f 8b magic (compress
// mode only), errors as "error: <msg>" on stderr with exit status 1.
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <string>
#include <vector>
#include <zlib.h>

#include "fqz.h"

static const char *kVersion = "dev-mi355x";

static void usage(FILE *f)
{
    fprintf(f,
            "fqpack - FASTQ block compressor (MI355X / HIP build)\n\n"
            "  fqpack [options] [-i in.fq] [-o out.fqz]     compress\n"
            "  fqpack -d [-i in.fqz] [-o out.fq]            decompress\n\n"
            "  -d          decompress mode\n"
            "  -i string   input file (default: stdin)\n"
            "  -o string   output file (default: stdout)\n"
            "  -c          write to stdout (compress mode)\n"
            "  -b uint     records per block (default 100000)\n"
            "  -w int      compression workers (default: NumCPU; the GPU pipeline ignores it)\n"
            "  -format n   container version to write: 2 (default, the reference's) or 3 (rANS-coded qualities; read by this tool only)\n"
            "  -index      with -format 3: append a table of the blocks (offset, records) behind the last block\n"
            "  -version    show version and exit\n"
            "  -h          show help\n");
}

static int fail(const std::string &msg)
{
    fprintf(stderr, "error: %s\n", msg.c_str());
    return 1;
}

// input side: a buffered FILE, optionally behind zlib's inflate (gzip members may be concatenated); the first two bytes
// are peeked to detect the gzip magic (cmd/fqpack/main.go:142-174)
struct Input {
    FILE *f = nullptr;
    bool gz = false, gz_end = false, failed = false;
    std::string err;
    z_stream zs;
    std::vector<uint8_t> ibuf;
    uint8_t peek[2];
    size_t n_peek = 0, peek_pos = 0;
    size_t raw_read(uint8_t *dst, size_t cap)
    {
        size_t got = 0;
        while (peek_pos < n_peek && got < cap) dst[got++] = peek[peek_pos++];
        if (got < cap) got += fread(dst + got, 1, cap - got, f);
        return got;
    }
};

static long input_read(void *u, uint8_t *dst, size_t cap)
{
    Input *in = (Input *)u;
    if (!in->gz) {
        size_t r = in->raw_read(dst, cap);
        if (ferror(in->f)) { in->failed = true; in->err = "read error"; return -1; }
        return (long)r;
    }
    size_t w = 0;
    while (w < cap && !in->gz_end) {
        if (!in->zs.avail_in) {
            size_t r = in->raw_read(in->ibuf.data(), in->ibuf.size());
            if (ferror(in->f)) { in->failed = true; in->err = "read error"; return -1; }
            in->zs.next_in = in->ibuf.data();
            in->zs.avail_in = (uInt)r;
            if (!r) { in->failed = true; in->err = "gzip: unexpected EOF"; return -1; }
        }
        in->zs.next_out = dst + w;
        in->zs.avail_out = (uInt)(cap - w > (1u << 30) ? (1u << 30) : cap - w);
        const uInt before = in->zs.avail_out;
        int rc = inflate(&in->zs, Z_NO_FLUSH);
        w += before - in->zs.avail_out;
        if (rc == Z_STREAM_END) { // another member may follow (concatenated gzip files)
            if (!in->zs.avail_in) {
                size_t r = in->raw_read(in->ibuf.data(), in->ibuf.size());
                in->zs.next_in = in->ibuf.data();
                in->zs.avail_in = (uInt)r;
            }
            if (in->zs.avail_in && inflateReset(&in->zs) == Z_OK) continue;
            in->gz_end = true;
        } else if (rc != Z_OK && rc != Z_BUF_ERROR) { in->failed = true; in->err = "gzip: invalid input"; return -1; }
    }
    return (long)w;
}

static int output_write(void *u, const uint8_t *src, size_t n) { return fwrite(src, 1, n, (FILE *)u) != n; }

int main(int argc, char **argv)
{
    bool decompress = false, to_stdout = false;
    std::string in_path, out_path;
    unsigned long block_size = FQZ_DEFAULT_BLOCK_SIZE;
    long workers = 0;
    unsigned long format = 0;
    bool index = false;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](const char *name) -> const char * {
            if (i + 1 >= argc) { fprintf(stderr, "flag needs an argument: %s\n", name); usage(stderr); exit(2); }
            return argv[++i];
        };
        if (a == "-d" || a == "--d") decompress = true;
        else if (a == "-c" || a == "--c") to_stdout = true;
        else if (a == "-h" || a == "--h" || a == "-help" || a == "--help") { usage(stderr); return 0; }
        else if (a == "-version" || a == "--version") { printf("fqpack version %s\n", kVersion); return 0; }
        else if (a == "-i" || a == "--i") in_path = need("-i");
        else if (a == "-o" || a == "--o") out_path = need("-o");
        else if (a == "-b" || a == "--b") block_size = strtoul(need("-b"), nullptr, 10);
        else if (a == "-w" || a == "--w") workers = strtol(need("-w"), nullptr, 10);
        else if (a == "-index" || a == "--index") index = true;
        else if (a == "-format" || a == "--format") format = strtoul(need("-format"), nullptr, 10); // (not a flag of the reference: SURVEY 8 f-4 asks for version 3 "behind a flag")
        else if (a == "--") { for (int j = i + 1; j < argc; j++) pos.push_back(argv[j]); break; }
        else if (a.size() > 1 && a[0] == '-') { fprintf(stderr, "flag provided but not defined: %s\n", a.c_str()); usage(stderr); return 2; }
        else { for (int j = i; j < argc; j++) pos.push_back(argv[j]); break; } // Go's flag package stops at the first positional
    }
    if (!pos.empty() && in_path.empty()) in_path = pos[0];
    if (pos.size() > 1 && out_path.empty()) out_path = pos[1];

    FILE *fin = stdin;
    if (!in_path.empty() && in_path != "-") {
        fin = fopen(in_path.c_str(), "rb");
        if (!fin) return fail("cannot open input: open " + in_path + ": " + strerror(errno));
    }
    Input in;
    in.f = fin;
    setvbuf(fin, nullptr, _IOFBF, 1 << 20);                               // 1 MiB buffered IO (main.go:123-188)
    in.n_peek = fread(in.peek, 1, 2, fin);
    if (ferror(fin)) return fail("cannot inspect input: read error");
    if (!decompress) {
        bool gz_name = in_path.size() >= 3 && !strcasecmp(in_path.c_str() + in_path.size() - 3, ".gz");
        bool gz_magic = in.n_peek == 2 && in.peek[0] == 0x1f && in.peek[1] == 0x8b;
        if (gz_name || gz_magic) {
            memset(&in.zs, 0, sizeof in.zs);
            if (inflateInit2(&in.zs, 15 + 16) != Z_OK) return fail("cannot open gzip input");
            in.gz = true;
            in.ibuf.resize(1 << 20);
        }
    }
    FILE *fout = stdout;
    if (!(out_path.empty() || out_path == "-" || to_stdout)) {
        fout = fopen(out_path.c_str(), "wb");
        if (!fout) return fail("cannot create output: open " + out_path + ": " + strerror(errno));
    }
    setvbuf(fout, nullptr, _IOFBF, 1 << 20);

    fqz_ctx *ctx = nullptr;
    int rc = fqz_ctx_create(0, &ctx);
    if (rc) return fail(std::string(fqz_strerror(rc)) + " (" + fqz_last_hip_error() + ")");
    if (decompress) {
        fqz_decompress_options o = {(int32_t)workers};
        rc = fqz_decompress_stream(ctx, input_read, &in, output_write, fout, &o);
    } else {
        fqz_options o = {(uint32_t)block_size, (int32_t)workers, (uint32_t)format, index ? 1u : 0u};
        rc = fqz_compress_stream(ctx, input_read, &in, output_write, fout, &o);
    }
    fqz_ctx_destroy(ctx);
    if (in.gz) inflateEnd(&in.zs);
    if (fin != stdin) fclose(fin);
    if (rc) {
        if (in.failed) return fail(in.err);
        const char *ctxmsg = decompress ? "" : (rc <= FQZ_E_HDR_AT && rc >= FQZ_E_LEN_MISMATCH ? "parsing FASTQ: " : "");
        return fail(std::string(ctxmsg) + fqz_strerror(rc));
    }
    if (fflush(fout)) return fail("write error");
    if (fout != stdout) fclose(fout);
    return 0;
}
