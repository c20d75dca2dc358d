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
            "  -version    show version and exit\n"
            "  -h          show help\n");
}

static int fail(const std::string &msg)
{
    fprintf(stderr, "error: %s\n", msg.c_str());
    return 1;
}

static bool read_all(FILE *f, std::vector<uint8_t> &buf)
{
    size_t n = 0;
    buf.resize(1 << 20);
    for (;;) {
        if (n == buf.size()) buf.resize(buf.size() * 2);
        size_t r = fread(buf.data() + n, 1, buf.size() - n, f);
        n += r;
        if (!r) break;
    }
    buf.resize(n);
    return !ferror(f);
}

static bool gunzip(const std::vector<uint8_t> &in, std::vector<uint8_t> &out, std::string &err)
{
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 16) != Z_OK) { err = "cannot open gzip input"; return false; }
    out.resize(in.size() * 4 + (1 << 20));
    zs.next_in = (Bytef *)in.data();
    zs.avail_in = (uInt)in.size();
    size_t w = 0, consumed = 0;
    for (;;) {
        if (w == out.size()) out.resize(out.size() * 2);
        size_t room = out.size() - w;
        zs.next_out = out.data() + w;
        zs.avail_out = (uInt)(room > (1u << 30) ? (1u << 30) : room);
        uInt before_out = zs.avail_out;
        if (!zs.avail_in && consumed < in.size()) {
            size_t left = in.size() - consumed;
            zs.next_in = (Bytef *)in.data() + consumed;
            zs.avail_in = (uInt)(left > (1u << 30) ? (1u << 30) : left);
        }
        uInt before_in = zs.avail_in;
        int rc = inflate(&zs, Z_NO_FLUSH);
        w += before_out - zs.avail_out;
        consumed += before_in - zs.avail_in;
        if (rc == Z_STREAM_END) {
            if (consumed < in.size() && inflateReset(&zs) == Z_OK) continue; // concatenated members
            break;
        }
        if (rc != Z_OK && rc != Z_BUF_ERROR) { inflateEnd(&zs); err = "gzip: invalid input"; return false; }
        if (rc == Z_BUF_ERROR && consumed >= in.size() && zs.avail_out) { inflateEnd(&zs); err = "gzip: unexpected EOF"; return false; }
    }
    inflateEnd(&zs);
    out.resize(w);
    return true;
}

int main(int argc, char **argv)
{
    bool decompress = false, to_stdout = false;
    std::string in_path, out_path;
    unsigned long block_size = FQZ_DEFAULT_BLOCK_SIZE;
    long workers = 0;
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
    std::vector<uint8_t> in, tmp, out;
    if (!read_all(fin, in)) return fail("cannot inspect input: read error");
    if (fin != stdin) fclose(fin);
    if (!decompress) {
        bool gz_name = in_path.size() >= 3 && !strcasecmp(in_path.c_str() + in_path.size() - 3, ".gz");
        bool gz_magic = in.size() >= 2 && in[0] == 0x1f && in[1] == 0x8b;
        if (gz_name || gz_magic) {
            std::string err;
            if (!gunzip(in, tmp, err)) return fail(err);
            in.swap(tmp);
        }
    }

    fqz_ctx *ctx = nullptr;
    int rc = fqz_ctx_create(0, &ctx);
    if (rc) return fail(std::string(fqz_strerror(rc)) + " (" + fqz_last_hip_error() + ")");
    size_t n = 0;
    if (decompress) {
        fqz_decompress_options o = {(int32_t)workers};
        rc = fqz_decompress(ctx, in.data(), in.size(), nullptr, 0, &n, &o);
        if (!rc) { out.resize(n ? n : 1); rc = fqz_decompress(ctx, in.data(), in.size(), out.data(), n, &n, &o); }
    } else {
        fqz_options o = {(uint32_t)block_size, (int32_t)workers};
        out.resize(fqz_encode_bound(in.size()) + FQZ_FILE_HEADER_SIZE);
        rc = fqz_compress(ctx, in.data(), in.size(), out.data(), out.size(), &n, &o);
    }
    fqz_ctx_destroy(ctx);
    if (rc) {
        const char *ctxmsg = decompress ? "" : (rc <= FQZ_E_HDR_AT && rc >= FQZ_E_LEN_MISMATCH ? "parsing FASTQ: " : "");
        return fail(std::string(ctxmsg) + fqz_strerror(rc));
    }
    FILE *fout = stdout;
    if (!(out_path.empty() || out_path == "-" || to_stdout)) {
        fout = fopen(out_path.c_str(), "wb");
        if (!fout) return fail("cannot create output: open " + out_path + ": " + strerror(errno));
    }
    if (n && fwrite(out.data(), 1, n, fout) != n) return fail("write error");
    if (fflush(fout)) return fail("write error");
    if (fout != stdout) fclose(fout);
    return 0;
}
