"""instruction histogram of one kernel of a device assembly file (hipcc -S --cuda-device-only): per opcode and per class, for the
whole function and for its hottest loop (the innermost basic-block range that holds the most v_mad_u64_u32).
usage: isa_hist.py file.s <substring of the mangled kernel name> [--loop]"""
import collections, re, sys
src = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
lab = re.compile(r"^([A-Za-z_.$][\w.$]*):")
start = next(i for i, l in enumerate(src) if lab.match(l) and pat in lab.match(l).group(1) and not l.startswith("."))
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
body = src[start + 1:end + 1]
def classify(op):
    if op.startswith("v_mad_u64_u32"): return "mul: v_mad_u64_u32"
    if op.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mad_u32_u24")): return "mul: other"
    if op.startswith(("v_lshl_add_u64", "v_lshrrev_b64", "v_lshlrev_b64", "v_add_co", "v_addc_co", "v_add_u64")): return "valu 64-bit (column joins, shifts, carries)"
    if op.startswith(("v_and", "v_or", "v_xor", "v_not", "v_bfe", "v_bfi", "v_lshrrev_b32", "v_lshlrev_b32", "v_alignbit", "v_and_or", "v_lshl_or", "v_lshl_add_u32", "v_or3", "v_xad", "v_perm")): return "valu 32-bit logic / shift (masks, pack, unpack)"
    if op.startswith(("v_add", "v_sub", "v_add3", "v_subrev")): return "valu 32-bit add / sub"
    if op.startswith(("v_cmp", "v_cndmask", "v_mov", "v_readfirstlane", "v_readlane", "v_writelane", "v_accvgpr", "v_swap")): return "valu compare / select / move"
    if op.startswith("v_"): return "valu other"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "s_waitcnt / s_nop"
    if op.startswith("s_"): return "salu / branch"
    return "other"
def hist(lines):
    ops, cls = collections.Counter(), collections.Counter()
    for l in lines:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or lab.match(l): continue
        op = t.split()[0]
        ops[op] += 1; cls[classify(op)] += 1
    return ops, cls
def show(title, lines):
    ops, cls = hist(lines)
    n = sum(ops.values()); v = sum(c for k, c in cls.items() if k.startswith(("mul", "valu")))
    print(f"## {title}: {n} instructions, {v} VALU")
    for k, c in sorted(cls.items(), key=lambda kv: -kv[1]): print(f"  {c:6d}  {k}")
    print("  top opcodes: " + ", ".join(f"{k} {c}" for k, c in ops.most_common(24)))
show(f"{lab.match(src[start]).group(1)[:100]} (whole kernel)", body)
# hottest loop: the backward branch whose range holds the most multiplier instructions
labels = {lab.match(l).group(1): i for i, l in enumerate(body) if lab.match(l)}
best = None
for i, l in enumerate(body):
    m = re.match(r"\s*s_c?branch\S*\s+(\S+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        rng = body[labels[m.group(1)]:i + 1]
        k = sum(1 for x in rng if "v_mad_u64_u32" in x)
        if best is None or k > best[0]: best = (k, labels[m.group(1)], i)
if best: show(f"hottest loop (lines {best[1]}..{best[2]} of the kernel)", body[best[1]:best[2] + 1])
