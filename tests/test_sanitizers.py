"""AddressSanitizer + UndefinedBehaviorSanitizer on the CPU-side native code (GPU sanitizers are
not available on the pool): the oracle and the synthetic generator through a C driver, and the
native FASTQ I/O library through its own self-checking executable."""
import os
import subprocess

import helpers as H

SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")


def test_oracle_and_generator_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "san_driver")
    subprocess.check_call(["gcc", *SAN, "-std=c11", "-o", exe, os.path.join(H.ROOT, "tests", "sanitize_driver.c"),
                           os.path.join(H.ROOT, "oracle", "bdx_oracle.c"),
                           os.path.join(H.ROOT, "biodemux.jl_amd", "csrc", "bdx_synth.c"), "-lm", "-lpthread"])
    out = subprocess.run([exe], env=ENV, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "sanitize driver ok" in out.stdout


IO_DRIVER = r'''
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
extern "C" {
struct bdx_fq_file;
int32_t bdx_fq_open(const char*, bdx_fq_file**); void bdx_fq_close(bdx_fq_file*);
int64_t bdx_fq_index(const bdx_fq_file*, int64_t, int64_t, int64_t*, int32_t*, int64_t*, int32_t);
void bdx_fq_pack(const bdx_fq_file*, const int64_t*, const int32_t*, int64_t, uint8_t*, int64_t*, int32_t);
int64_t bdx_fq_seq_bytes(const int32_t*, int64_t);
int32_t bdx_fq_demux_write(const bdx_fq_file*, const int64_t*, const int32_t*, int64_t, const int32_t*, int32_t,
                           const char* const*, const int32_t*, const int32_t*, int32_t, int32_t, int32_t);
}
int main(int argc, char** argv) {
    std::string dir = argv[1];
    for (int variant = 0; variant < 3; ++variant) {
        std::string path = dir + "/in" + std::to_string(variant) + ".fastq";
        FILE* f = fopen(path.c_str(), "wb");
        const int n = 5003;
        for (int i = 0; i < n; ++i) {
            int len = (i * 37) % 61;
            fprintf(f, "@r%d%s", i, variant == 1 ? "\r\n" : "\n");
            for (int k = 0; k < len; ++k) fputc("ACGT"[(i + k * 7) & 3], f);
            fprintf(f, "%s+%s", variant == 1 ? "\r\n" : "\n", variant == 1 ? "\r\n" : "\n");
            for (int k = 0; k < len; ++k) fputc('I', f);
            fputc('\n', f);
        }
        if (variant == 2) fputs("@trunc\nACG", f);
        fclose(f);
        bdx_fq_file* h;
        if (bdx_fq_open(path.c_str(), &h)) return 2;
        int64_t cur = 0, total = 0;
        for (;;) {
            const int64_t B = 777;
            std::vector<int64_t> off(4 * B); std::vector<int32_t> len(4 * B); int64_t nxt = 0;
            int64_t nr = bdx_fq_index(h, cur, B, off.data(), len.data(), &nxt, 4);
            if (nr == 0) break;
            cur = nxt; total += nr;
            std::vector<uint8_t> seq((size_t)bdx_fq_seq_bytes(len.data(), nr) + 1); std::vector<int64_t> so(nr + 1);
            bdx_fq_pack(h, off.data(), len.data(), nr, seq.data(), so.data(), 4);
            std::vector<int32_t> cls(nr), ks(nr), ke(nr);
            for (int64_t i = 0; i < nr; ++i) { cls[i] = (int32_t)(i % 5); ks[i] = (i % 3) ? (int32_t)(i % 7) : -1; ke[i] = (int32_t)(len[4 * i + 1] - (i % 4)); }
            std::string p0 = dir + "/o" + std::to_string(variant) + "_0.fastq", p1 = dir + "/o" + std::to_string(variant) + "_1.fastq.gz";
            std::string p2 = dir + "/o" + std::to_string(variant) + "_2.fastq", p3 = p2 + "x", p4 = p2 + "y";
            const char* paths[5] = {p0.c_str(), p1.c_str(), p2.c_str(), p3.c_str(), p4.c_str()};
            if (bdx_fq_demux_write(h, off.data(), len.data(), nr, cls.data(), 5, paths, ks.data(), ke.data(), 1, 0, 4)) return 3;
        }
        bdx_fq_close(h);
        if (total != n + (variant == 2 ? 1 : 0)) { fprintf(stderr, "record count %lld\n", (long long)total); return 4; }
    }
    printf("io driver ok\n");
    return 0;
}
'''


def test_native_io_under_asan_ubsan(tmp_path):
    src = tmp_path / "io_driver.cpp"
    src.write_text(IO_DRIVER)
    exe = str(tmp_path / "io_driver")
    subprocess.check_call(["g++", *SAN, "-std=c++17", "-pthread", "-o", exe, str(src),
                           os.path.join(H.ROOT, "biodemux.jl_amd", "csrc", "bdx_io.cpp"), "-lz"])
    out = subprocess.run([exe, str(tmp_path)], env=ENV, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "io driver ok" in out.stdout
