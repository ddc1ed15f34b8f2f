"""CPU-side code under AddressSanitizer + UndefinedBehaviorSanitizer (the reference's CI builds with
--enable-sanitize, contrib/jenkins.sh:53): the oracle restatement, the host table generator and the host-only parts of
the shim (TRXD packer) are compiled with -fsanitize=address,undefined and driven through their entry points.
GPU code cannot be sanitized on this pool; these are the translation units that run on the host."""
import os
import subprocess
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]


def run(cmd, **kw):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, **kw)


def test_oracle_and_table_generator_under_asan_ubsan(tmp_path):
    src = tmp_path / "drive.cpp"
    src.write_text(textwrap.dedent(r'''
        #include <cstdio>
        #include <cstdlib>
        #include <cstring>
        #include <vector>
        #include <complex>
        extern "C" {
        #include "trx_oracle.h"
        }
        #include "trx_tables.h"
        int main()
        {
            // host table generator of the product (csrc/trx_tables.cpp)
            std::vector<char> blob(sizeof(trx_tables));
            if (trx_tables_generate(reinterpret_cast<trx_tables *>(blob.data())) != 0) return 1;
            const trx_tables *t = reinterpret_cast<const trx_tables *>(blob.data());
            if (t->magic != TRX_TABLES_MAGIC || !(t->unit_ok & 1u)) return 2;
            // oracle: setup, a synthetic burst through detect + demod at both rates, the TRXD datagram, teardown
            if (orc_setup() != 1) return 3;
            unsigned s = 12345u;
            for (int sps = 1; sps <= 4; sps += 3) {
                const int L = sps == 4 ? 625 : 156;
                std::vector<orc_cf> x(L);
                for (orc_cf &v : x) {
                    s = s * 1103515245u + 12345u; const float re = (float)((int)(s >> 16) % 2000 - 1000);
                    s = s * 1103515245u + 12345u; const float im = (float)((int)(s >> 16) % 2000 - 1000);
                    v.re = re; v.im = im;
                }
                std::vector<float> soft(640);
                for (int type = 0; type <= 6; type++) {
                    orc_ebp ebp;
                    memset(&ebp, 0, sizeof(ebp));
                    (void)orc_detect_any_burst(x.data(), L, 9 /* invalid tsc too */, 4.0f, sps, type, 63, &ebp);
                    const int rc = orc_detect_any_burst(x.data(), L, 3, 4.0f, sps, type, 3, &ebp);
                    if (rc > 0) (void)orc_demod_any_burst(x.data(), L, rc, sps, &ebp, soft.data());
                }
                // the batched core on int16 input, every slot type incl. OFF and IDLE
                std::vector<int16_t> iq(16 * L * 2);
                for (int16_t &v : iq) { s = s * 1103515245u + 12345u; v = (int16_t)((int)(s >> 16) % 4000 - 2000); }
                std::vector<orc_burst_params> prm(16);
                for (int i = 0; i < 16; i++) { prm[i].type = i % 7; prm[i].tsc = i % 8; prm[i].max_toa = (i % 3) * 30 + 3; prm[i].reserved = 0; }
                std::vector<orc_burst_result> res(16);
                std::vector<float> so(16 * 148);
                orc_pull_batch(iq.data(), 16, L, sps, prm.data(), 4.0f, 32767.0, res.data(), so.data(), 148, 1);
            }
            float soft[444];
            for (int i = 0; i < 444; i++) soft[i] = (float)(i % 7) / 6.0f;
            unsigned char pkt[460];
            for (unsigned ver = 0; ver < 3; ver++)
                for (int idle = 0; idle < 2; idle++)
                    (void)orc_trxd_pack(pkt, ver, 2715647u, 7, 63.9, -1.75, idle, idle, 1, 5, -12.5f, soft, idle ? 0 : 444);
            puts("sanitized ok");
            return 0;
        }
    '''))
    exe = tmp_path / "drive"
    obj = tmp_path / "trx_oracle.o"
    r = run(["gcc", "-std=gnu11", "-ffp-contract=off"] + SAN + ["-c", os.path.join(ROOT, "oracle", "trx_oracle.c"), "-o", str(obj)])
    assert r.returncode == 0, r.stdout
    r = run(["g++", "-std=gnu++17"] + SAN + ["-I", os.path.join(ROOT, "oracle"), "-I", os.path.join(ROOT, "osmo_trx_amd", "csrc"),
             str(src), os.path.join(ROOT, "osmo_trx_amd", "csrc", "trx_tables.cpp"), str(obj), "-o", str(exe), "-lm"])
    assert r.returncode == 0, r.stdout
    r = run([str(exe)])
    assert r.returncode == 0 and "sanitized ok" in r.stdout, r.stdout
