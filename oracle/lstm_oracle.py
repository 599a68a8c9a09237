"""TEST INFRASTRUCTURE — CPU restatement (oracle) of the LSTM inside the reference's MELD classifier; the product never
imports this module (only tests/ do).

What it restates: `nn.LSTM(input_size=D_m, hidden_size=D_e, num_layers=4, bidirectional=True, dropout=p)` as constructed by
MELDLSTMModel.__init__ (/root/reference/model.py:527-533) and called on the padded batch in MELDLSTMModel.forward
(/root/reference/model.py:546: `emotions, hidden = self.lstm(U)` — no packing, no mask).  The arithmetic itself lives in
third-party PyTorch (torch.nn.LSTM, unpinned in the reference's requirements.txt; torch 2.10 here): per layer and direction,
gate rows in the order i, f, g, o,
    G_t = x_t W_ih^T + b_ih + h_{t-1} W_hh^T + b_hh;  i, f, o = sigmoid(.), g = tanh(.);  c_t = f c_{t-1} + i g;  h_t = o tanh(c_t)
the reverse direction walks t = S-1 .. 0, a layer's output is [h forward | h reverse], and in train mode nn.Dropout(p) is
applied to the output of every layer but the last.

Pinned by tests/test_lstm_cpu.py: bit-for-bit formulas against torch.nn.LSTM itself (eval mode, float64, 1e-12) and against the
reference-generated MELDLSTMModel fixture (tests/golden/dialogue_rnn.npz `meld/*`) with this LSTM swapped in for nn.LSTM.
Train mode: the inter-layer dropout masks follow the build's Philox contract (oracle/philox.py, site SITE_LSTM + layer, one
offset per dropout call) so that the HIP path can be compared mask for mask."""
import torch

from . import ganffn_oracle as O

SITE_LSTM = 64


def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """x (S, B, In) -> h (S, B, H) of one direction"""
    S, B, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    out = [None] * S
    xg = x @ w_ih.T + b_ih
    for t in (range(S - 1, -1, -1) if reverse else range(S)):
        G = xg[t] + h @ w_hh.T + b_hh
        i, f, g, o = torch.sigmoid(G[:, :H]), torch.sigmoid(G[:, H:2 * H]), torch.tanh(G[:, 2 * H:3 * H]), torch.sigmoid(G[:, 3 * H:])
        c = f * c + i * g
        h = o * torch.tanh(c)
        out[t] = h
    return torch.stack(out, 0)


def lstm_forward(x, P, num_layers, p_drop=0.0, rng=None, prefix="", offsets=None):
    """nn.LSTM(...).forward(x)[0].  P: dict of torch's parameter names (weight_ih_l0, ..., bias_hh_l3_reverse) -> tensors.
    rng: O.Rng (train mode draws the inter-layer masks); offsets: the Philox offset of each inter-layer dropout call
    (default rng.offset, rng.offset + 1, ...: the HIP path draws one offset per call, in layer order)."""
    h = x
    for l in range(num_layers):
        f = lstm_direction(h, P[prefix + "weight_ih_l%d" % l], P[prefix + "weight_hh_l%d" % l], P[prefix + "bias_ih_l%d" % l],
                           P[prefix + "bias_hh_l%d" % l], False)
        b = lstm_direction(h, P[prefix + "weight_ih_l%d_reverse" % l], P[prefix + "weight_hh_l%d_reverse" % l],
                           P[prefix + "bias_ih_l%d_reverse" % l], P[prefix + "bias_hh_l%d_reverse" % l], True)
        h = torch.cat((f, b), dim=2)
        if l + 1 < num_layers and rng is not None and rng.train and p_drop > 0.0:
            off = offsets[l] if offsets is not None else rng.offset + l
            h = O._drop(h, p_drop, SITE_LSTM + l, rng.at(off))
    return h
