"""LSTM recognition network of SuPAIR (host-side PyTorch-ROCm code).

API and parameter names of the reference's model/video_prediction/encoder.py:7-57
(`RnnStates`: LSTM(c*w*h -> 256) unrolled for num_obj steps on the SAME flattened frame, then
256 -> 50 -> 8).  Because the input is identical at every step, its projection through
W_ih is computed once (one (nT x 1024) @ (1024 x 1024) GEMM on rocBLAS/hipBLASLt) instead of
num_obj times; the recurrent part is num_obj small GEMMs, and all gate math between the GEMMs is
the fused HIP cell of csrc/lstm.hip (`ops.encoder_lstm`), forward and backward; behind the fc1 GEMM the head
(sigmoid, fc2, and in the backward their gradients and bias sums) is one pass each way (`ops.encoder_head`).
"""
import torch
import torch.nn as nn

from .. import ops


class RnnStates(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.c = config
        self.z_size = 4
        self.lstm_size = 256
        img_size = self.c.channels * self.c.width * self.c.height
        self.rnn = nn.LSTM(img_size, self.lstm_size)
        self.fc1 = nn.Linear(self.lstm_size, 50)
        self.fc2 = nn.Linear(50, 2 * self.z_size)
        nn.init.xavier_uniform_(self.fc1.weight)
        nn.init.xavier_uniform_(self.fc2.weight)
        nn.init.constant_(self.fc1.bias, 0.1)
        nn.init.constant_(self.fc2.bias, 0.1)

    def forward(self, frames):
        """frames (-1, c, w, h) -> (-1, num_obj, 8): per-object (mean, std) codes of [sx, sy/sx, x, y]."""
        x = frames.flatten(start_dim=1)
        rnn = self.rnn
        gemm = getattr(self.c, 'encoder_gemm', 'bf16x3')
        hs = ops.encoder_lstm(x, rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.bias_ih_l0, rnn.bias_hh_l0, self.c.num_obj,
                              time_major=True, gemm=gemm)     # (num_obj, n, 256), as the LSTM kernels write it
        fc1, fc2 = self.fc1, self.fc2
        # (n, num_obj, 8): the head writes its small output frame-major itself
        return ops.encoder_head(hs, fc1.weight, fc1.bias, fc2.weight, fc2.bias, gemm=gemm, step_major=True)
