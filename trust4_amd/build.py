"""Build trust4_amd/libt4hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "t4_api.hip")
SRC_HOST = os.path.join(HERE, "csrc", "t4_assembler.cpp")
OUT = os.path.join(HERE, "libt4hip.so")
DEPS = [SRC, SRC_HOST, os.path.join(HERE, "host", "trust4_main.cpp"), os.path.join(HERE, "host", "fastq_extractor_main.cpp"),
        os.path.join(HERE, "host", "bam_extractor_main.cpp"), os.path.join(HERE, "host", "bam_reader.h"), os.path.join(HERE, "host", "read_format.h"),
        os.path.join(HERE, "host", "seq_reader.h"), os.path.join(HERE, "host", "process_read.h"), os.path.join(HERE, "csrc", "t4_internal.h"),
        os.path.join(HERE, "csrc", "t4_kernels.h"), os.path.join(HERE, "csrc", "t4_wide.h"), os.path.join(HERE, "csrc", "t4_device.h"),
        os.path.join(os.path.dirname(HERE), "include", "trust4_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def build(force=False, verbose=False):
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", OUT, SRC, SRC_HOST, "-lz", "-lpthread", "-ldl"]   # RCCL (t4_comm) is bound by dlopen when a communicator is first asked for
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    build_driver(verbose)
    return OUT


def build_driver(verbose=False):
    """trust4_amd/bin/trust4-hip (stage 1, the reference's trust4 command line) and trust4_amd/bin/fastq-extractor-hip (stage-0
    candidate filter, the reference's fastq-extractor) and trust4_amd/bin/bam-extractor-hip (the same for BAM input, the reference's
    bam-extractor), linked against libt4hip.so."""
    out_dir = os.path.join(HERE, "bin")
    os.makedirs(out_dir, exist_ok=True)
    out = None
    for name, src in (("fastq-extractor-hip", "fastq_extractor_main.cpp"), ("bam-extractor-hip", "bam_extractor_main.cpp"), ("trust4-hip", "trust4_main.cpp")):
        out = os.path.join(out_dir, name)
        cmd = ["g++", "-O2", "-std=c++17", "-o", out, os.path.join(HERE, "host", src), "-L" + HERE, "-lt4hip", "-Wl,-rpath," + HERE,
               "-Wl,-rpath,$ORIGIN/..", "-lz", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
