"""Stock-PyTorch re-declaration of the six GAN-FFN networks (TEST INFRASTRUCTURE).

Purpose: (1) the `cpu_baseline` leg of bench.py — the reference's .py files do not
travel to the GPU box, so the host-CPU baseline the speed-up is quoted against is
this file's stock `nn.TransformerEncoder` stack run with the reference's sub-step
logic; (2) a second, independent check of oracle/ganffn_oracle.py.

One parametrised class replaces the reference's six near-identical ones
(/root/reference/model.py:1200-1397); `state_dict()` keys and shapes are the
reference's (position_encoding.pe, encoder_layer.*, transformer_encoder.layers.N.*,
fc1, fc2[, fc3, object]) so weights are interchangeable — tests/test_oracle_golden.py
verifies that against the golden fixtures.
"""
import math

import torch
import torch.nn as nn

SPECS = {
    # name: (kind, d_model, nhead, fc dims, has_object)
    "AcousticGenerator": ("gen", 100, 10, (512,), False),
    "TextGenerator": ("gen", 100, 10, (512,), False),
    "VisualGenerator": ("gen", 512, 8, (1024,), False),
    "AcousticDiscriminator": ("disc", None, 10, (64, 16), False),
    "TextDiscriminator": ("disc", None, 10, (64, 16), False),
    "VisualDiscriminator": ("disc", None, 10, (64, 16), True),
    # extension (no reference GAN path for MELD): the same recipe at MELD's feature widths, text 600 / audio 300
    # (train_MELD.py:143, dataloader.py:93-95); has_object = the raw-modality width `object` maps to D_h
    "MELDTextGenerator": ("gen", 600, 10, (1024,), False),
    "MELDAudioGenerator": ("gen", 300, 10, (512,), False),
    "MELDTextDiscriminator": ("disc", None, 10, (64, 16), 600),
    "MELDAudioDiscriminator": ("disc", None, 10, (64, 16), 300),
}


class _PE(nn.Module):
    def __init__(self, d_model, p=0.2, max_len=110):
        super().__init__()
        self.dropout = nn.Dropout(p)
        pos = torch.arange(max_len).unsqueeze(1)
        div = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, 1, d_model)
        pe[:, 0, 0::2] = torch.sin(pos * div)
        pe[:, 0, 1::2] = torch.cos(pos * div)
        self.register_buffer("pe", pe)

    def forward(self, x):
        return self.dropout(x + self.pe[: x.size(0)])


class StockNet(nn.Module):
    """kind 'gen': enc -> gelu -> drop -> gelu(drop(fc1)) -> gelu(drop(fc2)).
    kind 'disc': [object] -> enc -> gelu -> gelu(drop(fc1)) -> gelu(drop(fc2)) -> sigmoid(drop(fc3))."""

    def __init__(self, name, D_h=100, dropout=0.2, num_layers=8):
        super().__init__()
        kind, d_model, nhead, fcs, has_obj = SPECS[name]
        d_model = d_model or D_h
        self.kind = kind
        self.position_encoding = _PE(d_model)
        self.encoder_layer = nn.TransformerEncoderLayer(d_model=d_model, nhead=nhead)
        self.transformer_encoder = nn.TransformerEncoder(encoder_layer=self.encoder_layer, num_layers=num_layers)
        self.obj_in = (512 if has_obj is True else int(has_obj)) if has_obj else 0
        if has_obj:
            self.object = nn.Linear(self.obj_in, 100)
        if kind == "gen":
            self.fc1 = nn.Linear(d_model, fcs[0])
            self.fc2 = nn.Linear(fcs[0], D_h)
        else:
            self.fc1 = nn.Linear(d_model, fcs[0])
            self.fc2 = nn.Linear(fcs[0], fcs[1])
            self.fc3 = nn.Linear(fcs[1], 1)
        self.gelu = nn.GELU()
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        if hasattr(self, "object") and x.size(-1) == self.obj_in:
            x = self.object(x)
        t = self.gelu(self.transformer_encoder(self.position_encoding(x)))
        if self.kind == "gen":
            t = self.dropout(t)
            t = self.gelu(self.dropout(self.fc1(t)))
            return self.gelu(self.dropout(self.fc2(t)))
        t = self.gelu(self.dropout(self.fc1(t)))
        t = self.gelu(self.dropout(self.fc2(t)))
        return torch.sigmoid(self.dropout(self.fc3(t)))


def stock_train_disc(disc, real_d, gen, real_g, opt, bce, valid, fake):
    """Sub-step semantics of /root/reference/train_IEMOCAP.py:213-226."""
    disc.train(); gen.eval()
    opt.zero_grad()
    loss = (bce(disc(real_d), valid) + bce(disc(gen(real_g).detach()), fake)) / 2.0
    res = loss.detach().cpu().numpy()
    loss.backward()
    opt.step()
    return res


def stock_train_gen(gen, real_g, disc, opt, bce, valid, fake):
    """Sub-step semantics of /root/reference/train_IEMOCAP.py:242-251."""
    gen.train(); disc.eval()
    opt.zero_grad()
    loss = bce(disc(gen(real_g)), valid)
    res = loss.detach().cpu().numpy()
    loss.backward()
    opt.step()
    return res


def stock_gan_iteration(gens, discs, opts, batch, schedule):
    """One batch of the 12-sub-step schedule on stock modules (CPU baseline body)."""
    S, B = batch["text"].shape[:2]
    valid = torch.ones(S, B, 1)
    fake = torch.zeros(S, B, 1)
    bce = nn.BCELoss()
    out = {}
    for kind, who, partner in schedule:
        if kind == "D":
            out["%s_D_loss" % who] = stock_train_disc(discs[who], batch[who], gens[partner], batch[partner],
                                                      opts[("D", who)], bce, valid, fake)
        else:
            out["%s_G_loss" % who] = stock_train_gen(gens[who], batch[who], discs[partner],
                                                     opts[("G", who)], bce, valid, fake)
    return out


IEMOCAP_NETS = {"G": {"acoustic": "AcousticGenerator", "visual": "VisualGenerator", "text": "TextGenerator"},
                "D": {"acoustic": "AcousticDiscriminator", "visual": "VisualDiscriminator", "text": "TextDiscriminator"}}
MELD_NETS = {"G": {"acoustic": "MELDAudioGenerator", "text": "MELDTextGenerator"},
             "D": {"acoustic": "MELDAudioDiscriminator", "text": "MELDTextDiscriminator"}}


def build_stock(D_h=100, dropout=0.2, lr=1e-4, b1=0.5, b2=0.6, num_layers=8, nets=None):
    nets = nets or IEMOCAP_NETS
    gens = {m: StockNet(n, D_h, dropout, num_layers) for m, n in nets["G"].items()}
    discs = {m: StockNet(n, D_h, dropout, num_layers) for m, n in nets["D"].items()}
    A = torch.optim.Adam
    opts = {}
    for m in gens:          # train_IEMOCAP.py:292-297: G lr, text-G 1.1 lr, every D lr / 2
        opts[("G", m)] = A(gens[m].parameters(), lr=lr * 1.1 if m == "text" else lr, betas=(b1, b2))
    for m in discs:
        opts[("D", m)] = A(discs[m].parameters(), lr=lr / 2, betas=(b1, b2))
    return gens, discs, opts
