"""Where loop C's resident kernel spends its time: s_memtime ticks per wave slot and phase.

    python tools/phase_clocks.py build          # here (no GPU): instrumented build -> tools/ubench/bin/libarah_clk.so
    python tools/phase_clocks.py run [frames]   # on the GPU box

The instrumented library is the product source compiled with -DARAH_CLOCKS (k_canon_solve accumulates the ticks
between consecutive marks per wave; the marks cost ~11 % of the wave cycles themselves, MI355X_MICROARCH.md).
Phases: 0 refill, 1 barrier after refill, 2 input layer, 3/5/7 GEMM of hidden layer 1..3, 4/6/8 barrier + Softplus
epilogue of that layer, 9 output layer, 10 per-point tail (softmax tree, blend, Broyden), 11 closing barrier.
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tools", "ubench", "bin", "libarah_clk.so")
NAMES_TILE = ["refill", "bar0", "layer0", "gemm1", "epi1", "gemm2", "epi2", "gemm3", "epi3", "out", "tail", "bar1"]
# k_canon_wave (the default): 0 refill + claim, 1 first chunk of the input layer, 2..4 hidden layers 1..3 (MFMA steps with the
# epilogue parts that ride along), 5 output layer, 6 per-point tail
NAMES_WAVE = ["refill", "layer0.0", "layer1", "layer2", "layer3", "out", "tail"]
NAMES_SHADE = ["load", "trunk", "head", "sweep", "colin", "colour", "store", "tr.gemm", "tr.epi", "tr.bar", "sw.gemm", "sw.epi", "sw.bar"]
NAMES = NAMES_TILE if os.environ.get("ARAH_CANON_KERNEL") == "tile" else NAMES_WAVE


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    csrc = os.path.join(ROOT, "arah_release_amd", "csrc")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-fno-slp-vectorize", "-DARAH_CLOCKS",
                    "-shared", "-fPIC", os.path.join(csrc, "arah_hip.hip"), "-o", LIB], check=True, cwd=csrc)


def run(frames):
    os.environ["ARAH_LIB_PATH"] = LIB
    sys.path.insert(0, ROOT)
    import torch
    from arah_release_amd import config, hip, synthetic
    dev = torch.device("cuda", 0)
    model, _ = config.build_synthetic_model("zju377_mono", 64, 16, 16, device=dev)
    scene = synthetic.SyntheticScene(0)
    lib = hip.load_library()
    tracer = model.idhr_network.ray_tracer
    with torch.no_grad():
        model(scene.make_inputs(512, 512, frame_idx=0, device=dev), eval=True)   # warm-up, sizes the workspace
        ws = tracer.workspace(dev)
        ws.reset_counters()
        for f in range(frames):
            model(scene.make_inputs(512, 512, frame_idx=1 + f, device=dev), eval=True)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 256)()
    rc = lib.arah_debug_clocks(C.c_void_p(ws.buf.data_ptr()), out, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    ctr = ws.counters()
    print("frames %d, skinning evaluations %d" % (frames, ctr["n_skin_fwd"]))
    print("%-8s" % "phase" + "".join("%9s" % ("wave%d" % w) for w in range(8)) + "   (% of the wave's ticks)")
    tot = [sum(out[w * 16 + i] for i in range(16)) for w in range(8)]
    for i, nm in enumerate(NAMES):
        print("%-8s" % nm + "".join("%9.1f" % (100.0 * out[w * 16 + i] / max(tot[w], 1)) for w in range(8)))
    print("%-8s" % "Gticks" + "".join("%9.2f" % (tot[w] / 1e9) for w in range(8)))
    # k_shade (all of it with ARAH_FULL_SHADING=1): the second block of clocks
    sh = out[128:]
    tot = [sum(sh[w * 16 + i] for i in range(16)) for w in range(8)]
    if sum(tot):
        print("k_shade, shaded samples %d" % ctr["n_col"])
        for i, nm in enumerate(NAMES_SHADE):
            print("%-8s" % nm + "".join("%9.1f" % (100.0 * sh[w * 16 + i] / max(tot[w], 1)) for w in range(8)))
        print("%-8s" % "Gticks" + "".join("%9.2f" % (tot[w] / 1e9) for w in range(8)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
