"""Prints the headline numbers and the per-kernel table of a bench.py JSON line.  usage: python tools/show_bench.py file.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
print(f"{d['value']:.4e} {d['unit']}  {d['ms_per_step']:.4f} ms/step  snr_min {d.get('parity', {}).get('snr_db_min')}")
r = d.get("roofline", {})
print("dominant", r.get("kernel"), "frac", r.get("frac"), "whole-step frac", r.get("whole_step", {}).get("frac_mfma"))
for k, v in list(r.get("by_kernel", {}).items())[:12]:
    print(f"  {k:40s} {v['ms_per_step']:.4f} ms  {v['tflops']:7.1f} TF  x{v['launches']}")
print("speaker_encoder", d.get("speaker_encoder"))
