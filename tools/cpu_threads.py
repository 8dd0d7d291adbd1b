#!/usr/bin/env python3
"""How many host threads should the CPU baseline use?  (development tool)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from oracle import oracle as O
from ppt_amd import weights as W
from ppt_amd.models import ULIP_models as M
names = M.dataset_classnames("modelnet40"); ids, nl = M.tokenize_prompts(names, 32); eot = ids.argmax(-1).numpy()
sd = W.ulip_pointbert_state_dict(0); emb = W.synth_prompt_embedding(40, 0)
pc, start = W.synth_clouds(8, 1024, seed=1234); pc = torch.from_numpy(pc); lab = torch.zeros(8, dtype=torch.long)
print("cpu_count", os.cpu_count())
for nt in (8, 16, 32, 64):
    torch.set_num_threads(nt)
    t0 = time.time(); O.train_step(sd, pc, lab, start, emb, nl, eot); t1 = time.time(); O.train_step(sd, pc, lab, start, emb, nl, eot); t2 = time.time()
    print(nt, "threads: warm", round(t1 - t0, 2), "s, step", round(t2 - t1, 2), "s", flush=True)
