#!/usr/bin/env python3
"""Instruction mix and register / LDS use of kernels in a `hipcc -S --offload-device-only` listing.

    python scripts/isa_stats.py <file.s> <substring of the mangled kernel name> [...]
"""
import collections
import re
import sys

KEYS = ["s_load_dword", "s_load_dwordx2", "s_load_dwordx4", "global_load_dword", "global_load_dwordx2", "global_load_dwordx3",
        "global_load_dwordx4", "global_load_ubyte", "buffer_load_dword", "buffer_load_dwordx2", "global_store_dword",
        "global_store_dwordx2", "ds_read_b64", "ds_read_b32", "ds_write_b64", "ds_write_b32", "s_waitcnt", "s_barrier",
        "scratch_load_dword", "scratch_store_dword"]


def kernels(text):
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s+s_endpgm", text, re.S | re.M):
        yield m.group(1), m.group(2)


def main():
    text = open(sys.argv[1]).read()
    wants = sys.argv[2:]
    for name, body in kernels(text):
        if wants and not any(w in name for w in wants):
            continue
        ops = [l.split()[0] for l in body.splitlines()
               if l.strip() and not l.strip().startswith((".", ";")) and not l.strip().endswith(":")]
        c = collections.Counter(ops)
        meta = text[text.index(".amdhsa_kernel " + name):]
        meta = meta[:meta.index(".end_amdhsa_kernel")]
        get = lambda k: (re.search(r"\.amdhsa_" + k + r" (\d+)", meta) or [None, "?"])[1]
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        salu = sum(v for k, v in c.items() if k.startswith("s_") and not k.startswith(("s_load", "s_waitcnt", "s_buffer_load")))
        print(f"{name}\n   instructions {len(ops)}  VALU {valu}  SALU {salu}  vgpr {get('next_free_vgpr')}  sgpr {get('next_free_sgpr')}"
              f"  accum_offset {get('accum_offset')}  lds {get('group_segment_fixed_size')}  scratch {get('private_segment_fixed_size')}")
        print("   " + "  ".join(f"{k}={c[k]}" for k in KEYS if c.get(k)))


if __name__ == "__main__":
    main()
