#!/usr/bin/env python3
"""Static VALU instruction mix of the hot kernels, priced with the issue costs measured on this pool
(scripts/micro/valu_classes.hip -> profiles/r03_valu_instruction_classes.txt): what wave64-instruction rate can a kernel with THIS
mix of opcodes reach when every SIMD issues VALU instructions back to back?

    python scripts/valu_mix.py            -> profiles/valu_mix.json

The kernels are compiled to assembly with the library's own flags (CPU only, hipcc cross-compiles); an instruction counts when it
sits inside a loop of its kernel (the walk / path loops are where the time goes), every loop instruction with weight 1 -- a STATIC
mix, not weighted by how often each block runs.  bench.py reports the resulting rate beside the spec ceiling as
`attainable_for_mix`."""
import json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -fno-fast-math".split()

# cycles per wave64 instruction per SIMD at 2.4 GHz with 8 waves per SIMD (profiles/r03_valu_instruction_classes.txt)
FAST = 3.0          # mov, not, right shifts, and / or / xor, add / sub / mul / fma / fmac f32, add / sub u32
NORMAL = 4.7        # everything else that is not transcendental (min / max, compares, selects, conversions, left shifts, 3-operand
                    # integer ops, packed fp32, all fp64 arithmetic, lane reads)
TRANS = 8.9         # rcp, rsq, sqrt, exp, log, sin, cos
FAST_OPS = {"v_mov_b32", "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_f32", "v_sub_f32",
            "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32"}
TRANS_OPS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"}

KERNELS = {           # bench.py's kernel classes -> (file, mangled-name fragments of the builds that run in the bench workloads)
    "k_trace": ("trace.hip", ["7k_traceILb0ELb0ELb0E", "7k_traceILb0ELb0ELb1E"]),
    "k_shade_a": ("integrate.hip", ["9k_shade_aILb0E"]),
    "k_shade_b": ("integrate.hip", ["9k_shade_bILb0E"]),
    "k_tail": ("integrate.hip", ["6k_tailILi4E", "6k_tailILi3E"]),
}


def asm_of(src):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-S", "--cuda-device-only", os.path.join(ROOT, "pim_amd", "csrc", src), "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    text = open(out).read(); os.unlink(out)
    return text


def kernel_body(text, fragment):
    m = re.search(r"^(_Z\w*%s\w*):" % re.escape(fragment), text, re.M)
    if not m:
        return None
    end = text.index("s_endpgm", m.end())
    return text[m.end():end]


def mix(body):
    fast = normal = trans = 0
    in_loop = False
    for line in body.splitlines():
        if re.match(r"^\.LBB\d+_\d+:", line) or line.startswith("; %bb."):
            in_loop = "in Loop" in line or "Loop Header" in line
            continue
        m = re.match(r"^\s+(v_[a-z0-9_]+)", line)
        if not m or not in_loop:
            continue
        op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", m.group(1))
        if op in FAST_OPS: fast += 1
        elif op in TRANS_OPS: trans += 1
        else: normal += 1
    return fast, normal, trans


def main():
    texts = {}
    out = {"_source": "scripts/valu_mix.py: static mix of the VALU instructions inside loops, priced at %.1f / %.1f / %.1f cycles per wave64 instruction per SIMD "
                      "(profiles/r03_valu_instruction_classes.txt)" % (FAST, NORMAL, TRANS)}
    for name, (src, frags) in KERNELS.items():
        if src not in texts:
            texts[src] = asm_of(src)
        f = n = t = 0
        used = []
        for frag in frags:
            body = kernel_body(texts[src], frag)
            if body is None:
                continue
            a, b, c = mix(body); f += a; n += b; t += c; used.append(frag)
        total = f + n + t
        if not total:
            continue
        cycles = (f * FAST + n * NORMAL + t * TRANS) / total
        out[name] = {"builds": used, "loop_valu_instructions": total, "fast_fraction": f / total, "normal_fraction": n / total, "transcendental_fraction": t / total,
                     "cycles_per_instruction": cycles, "attainable_Ginst_per_s": 256 * 4 * 2.4 / cycles}
        print("%-10s %5d loop VALU instructions: fast %.2f normal %.2f transcendental %.3f -> %.2f cycles -> %.0f G wave-instructions/s" % (
            name, total, f / total, n / total, t / total, cycles, out[name]["attainable_Ginst_per_s"]))
    json.dump(out, open(os.path.join(ROOT, "profiles", "valu_mix.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
