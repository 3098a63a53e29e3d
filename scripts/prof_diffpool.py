#!/usr/bin/env python3
"""DiffPool cfg5 step, 20 eager iterations, for rocprofv3 --kernel-trace --stats (launch inventory)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import dense_encoders as E, synthetic, message_passing as mp
dev = torch.device("cuda"); torch.manual_seed(0)


class A:
    bias = True
hb5 = synthetic.host_batch(4, 16, "DD", 512)
g5, x5, lab5 = synthetic.to_device(hb5, dev)
dpm = E.SoftPoolingGcnEncoder(512, 89, 64, 64, 2, 3, 64, assign_ratio=0.125, num_pooling=2, bn=True, linkpred=False, args=A(),
                              assign_input_dim=89, final_dim="number_classes").to(dev)
for _ in range(20):
    dpm.zero_grad(set_to_none=True)
    with mp.deferred_loss():          # the cross-entropy inside the head's backward, as FlatTrainer(defer_loss=True) runs it
        dpm.loss(dpm(x5, g5, hb5["sizes"], assign_x=x5)[1], lab5).backward(gradient=mp.unit_seed(dev))
torch.cuda.synchronize()
