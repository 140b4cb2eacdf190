import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import audiocodec_amd
N=2048; B,K,C=256,234,2
codec=audiocodec_amd.AudioCodec(48000,N)
ws=codec.workspace(B,K,C)
r=ws.report
print(r["chunks_probed"], r["chosen_chunk"], [round(v,3) for v in r["encode_ms_by_chunk"]])
