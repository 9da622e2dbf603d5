"""Builds build_variants/libsdp_trace.so: the forward kernel with cycle stamps around the phases of a chunk
(each stamp drains all outstanding memory operations, so phases are serialised -- a breakdown, not a timing)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepblast_amd import build
p = os.path.join(ROOT, "deepblast_amd", "csrc", "sdp_kernels.hip")
orig = open(p).read()
s = orig
def ins_before(marker, code):
    global s
    assert marker in s, marker
    s = s.replace(marker, code + marker, 1)
def ins_after(marker, code):
    global s
    assert marker in s, marker
    s = s.replace(marker, marker + code, 1)
TS = "__builtin_amdgcn_s_waitcnt(0); tnow = __builtin_readcyclecounter(); "
ins_after("        for (int ci = 0; ci < nchunks; ++ci) {\n", "            unsigned long long tnow, tprev; " + TS + "tprev = tnow;\n#define PH(i) { " + TS + "ph[i] += tnow - tprev; tprev = tnow; }\n")
ins_before("        for (int ci = 0; ci < nchunks; ++ci) {\n", "        unsigned long long ph[8] = {0,0,0,0,0,0,0,0};\n")
ins_after("            load_block(bb_new);\n", "            PH(0)\n")
ins_before("            // ---- staged inputs of this chunk: one burst of LDS reads", "            PH(1)\n")
ins_before("            const bool interior = chunk_interior(c);", "            PH(2)\n")
ins_before("            // ---- publish K boundary values for the next strip", "            if (wf_done) ph[7] += 1;\n            PH(3)\n")
ins_before("            // ---- flush: one K-column aligned block per row", "            PH(4)\n")
ins_before("            if (more) write_block(bb_new);", "            PH(5)\n")
ins_after("            if (more) write_block(bb_new);\n", "            PH(6)\n")
ins_before("        if constexpr (!REV) {\n            if (t_final >= 0) {", "        if (PASS == PASS_FWD && lane == 0) { for (int i = 0; i < 8; ++i) p.vout[(b * 8 + s) * 8 + i] = (float)ph[i]; }\n")
s = s.replace("                    p.vout[b] = (float)((double)(int)hi32(vt_keep)", "                    if (false) p.vout[b] = (float)((double)(int)hi32(vt_keep)")
try:
    open(p, "w").write(s)
    build.build(out=os.path.join(ROOT, "build_variants", "libsdp_trace.so"), extra=sys.argv[1:])
finally:
    open(p, "w").write(orig)
