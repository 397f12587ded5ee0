"""Extracts one kernel instantiation from the ISA listing made by `make asm` and prints its instruction mix."""
import re, sys, os
d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "raytracer_amd", "csrc", "_asm")
tag = sys.argv[1] if len(sys.argv) > 1 else "ILi24ELb0ELb0ELi17E"
s = open(os.path.join(d, "rt_hip-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
m = re.search(r"^(_ZN\S*render_persistent_kernel" + tag + r"\S*):", s, re.M)
body = s[m.start():s.index(".end_amdhsa_kernel", m.end())]
open(os.path.join(d, "kB.s"), "w").write(body)
keys = ("v_div_fixup_f32 v_div_scale_f32 v_rcp_f32 v_sqrt_f32 v_rsq_f32 s_setreg s_denorm v_pk_fma_f32 v_pk_mul_f32 "
        "v_pk_add_f32 v_fma_f32 v_fmac_f32 s_waitcnt ds_read ds_write global_load buffer_load scratch_ v_cndmask s_cbranch "
        "v_mul_f32 v_add_f32 v_sub_f32 v_min v_max v_med3").split()
print(" ".join("%s=%d" % (k, body.count(k)) for k in keys))
print("instruction lines", len(re.findall(r"^\s+[vsgdb]\w*_\w+", body, re.M)))
