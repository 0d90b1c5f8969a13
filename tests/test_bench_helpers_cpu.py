"""bench.py's record helpers that need no GPU: kernel-name normalisation and the committed rocprofv3 summaries it reads
(profiles/rNN_kernel_trace_*.txt, written by tools/trace_summary.py)."""
import glob
import importlib.util
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(REPO, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_short_kernel_names():
    assert bench.short_kernel("void s8r4::k_fb_split8<0>(unsigned long long, unsigned long long, s8r4::FbSplitArgs)") == "k_fb_split8<0>"
    assert bench.short_kernel("s8r4::k_fb_split8(unsigned long long, unsign") == "k_fb_split8"       # summaries from before round 6
    assert bench.short_kernel("k_gemm_lds_adam(unsigned long long, unsigned long long, GemmGroup, AdamFuse)") == "k_gemm_lds_adam"
    assert bench.short_kernel("void k_gather_fused2<4, false>(FusedSampleArgs)") == "k_gather_fused2<4, false>"
    assert bench.short_kernel("  k_cycle_open  ") == "k_cycle_open"


def test_committed_summaries_parse_and_carry_a_fingerprint():
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r06_kernel_trace_b*_k*.txt")))
    assert files, "no round-6 kernel traces committed"
    for path in files:
        avg, sha = bench.committed_kernel_averages(path)
        assert sha and len(sha) == 16, path
        assert avg and all(v > 0 for v in avg.values()), path
        own = [k for k in avg if k.startswith("k_")]            # (torch's and the runtime's kernels keep whatever their templates leave)
        assert own and not any("[prologue" in k or " " in k for k in own), (path, own)     # the prologue row never stands in for the update's launch
    avg, _ = bench.committed_kernel_averages(os.path.join(REPO, "profiles", "r06_kernel_trace_b256_k4.txt"))
    assert 20 < avg["k_fb_split8<0>"] < 40 and 4 < avg["k_gemm_lds_adam"] < 15
    dp, _ = bench.committed_kernel_averages(os.path.join(REPO, "profiles", "r06_kernel_trace_b256_k4_forced_dp_peer.txt"))
    assert "k_fb_split8<1>" in dp and "k_gemm_lds_adam_peer" in dp and "k_fb_split8<0>" not in dp and "k_fb_slab8" not in dp
    rc, _ = bench.committed_kernel_averages(os.path.join(REPO, "profiles", "r06_kernel_trace_b256_k4_forced_dp_rccl.txt"))
    assert "k_fb_split8<2>" in rc and "k_gemm_lds" in rc and "k_adam_frag4" in rc
    old, sha5 = bench.committed_kernel_averages(os.path.join(REPO, "profiles", "r05_kernel_trace_b256_k4.txt"))
    assert "k_fb_split8" in old and sha5                        # the long form of earlier rounds still reads


def test_missing_summary_is_empty_not_an_error():
    assert bench.committed_kernel_averages(os.path.join(REPO, "profiles", "no_such_file.txt")) == ({}, None)
