"""Three-phase TimeGAN training steps, mirroring the loop bodies of
outdated_files/TimeGANv1.py:573-812 (embedding_trainer, supervisor_trainer, joint_trainer), with
optional data parallelism (one process per GPU, gradients summed over ranks with RCCL).

A "step" here is the body of the reference's `for idx, real_data in enumerate(dl)` loop for one
batch; the plotting / wandb block of the joint phase (TimeGANv1.py:814-896) is not part of it.
"""
from typing import Dict, List, Optional, Sequence

import torch
from torch.optim import Adam

from . import losses
from .modules import build_networks
from .parallel import all_reduce_grads, broadcast_parameters, world_size


class TimeGANTrainer:
    """Owns the five networks and the six Adam optimisers of TimeGANv1.py:926-938.

    The embedder has TWO optimisers, one used by the embedding phase and one by the joint phase
    (`emb_opt_side` / `emb_opt_main`, TimeGANv1.py:933-934), each with its own moment estimates.
    """

    def __init__(self, cfg: Dict, device=None, rnn_cls=None, group=None, nets=None):
        self.cfg = cfg
        self.device = torch.device(device if device is not None else cfg["system"]["device"])
        kw = {} if rnn_cls is None else {"rnn_cls": rnn_cls}
        self.nets = nets if nets is not None else build_networks(cfg, **kw)
        for k in self.nets:
            self.nets[k] = self.nets[k].to(self.device)
        self.group = group
        broadcast_parameters(self.nets.values(), group=group)
        lr = float(cfg["system"]["lr"])
        n = self.nets
        self.opt = {"emb_side": Adam(n["emb"].parameters(), lr=lr), "emb_main": Adam(n["emb"].parameters(), lr=lr),
                    "rec": Adam(n["rec"].parameters(), lr=lr), "sup": Adam(n["sup"].parameters(), lr=lr),
                    "g": Adam(n["g"].parameters(), lr=lr), "d": Adam(n["d"].parameters(), lr=lr)}
        self.d_threshold = float(cfg["d"]["threshold"])
        self.seq_len = int(cfg["system"]["seq_len"])

    # -- helpers ---------------------------------------------------------------------------------
    def _zero(self, *names):
        for k in names:
            self.nets[k].zero_grad()

    def _sync_grads(self, *names):
        all_reduce_grads([self.nets[k] for k in names], group=self.group)

    def lengths_of(self, x: torch.Tensor) -> List[int]:
        """What Energy.extract_time returns for a batch (data/Energy.py:12-19): every length = T."""
        return [x.shape[1]] * x.shape[0]

    # -- phase 1: embedding / recovery autoencoder (TimeGANv1.py:585-608) ---------------------------
    def embedding_step(self, x: torch.Tensor, t=None) -> float:
        n = self.nets
        t = self.lengths_of(x) if t is None else t
        self._zero("emb", "rec", "sup")
        e_loss0, _ = losses.embedding_forward_side(n["emb"], n["rec"], x, t, group=self.group)
        e_loss0.backward()
        self._sync_grads("emb", "rec")
        self.opt["emb_side"].step()
        self.opt["rec"].step()
        return float(e_loss0.detach())

    # -- phase 2: supervisor (TimeGANv1.py:662-683) -------------------------------------------------
    def supervisor_step(self, x: torch.Tensor, t=None) -> float:
        n = self.nets
        t = self.lengths_of(x) if t is None else t
        self._zero("emb", "sup")
        s_loss, _, _ = losses.supervisor_forward(n["emb"], n["sup"], x, t, group=self.group)
        s_loss.backward()
        self._sync_grads("sup")
        self.opt["sup"].step()
        return float(s_loss.detach())

    # -- phase 3: joint adversarial training (TimeGANv1.py:750-812) --------------------------------
    def joint_step(self, x: torch.Tensor, t=None, noise: Optional[Sequence[torch.Tensor]] = None) -> Dict:
        """One joint iteration: 10 x (generator step + embedder step) and one discriminator step.

        `noise`: optional 11 tensors shaped like x used in place of torch.rand_like(x)
        (TimeGANv1.py:762,793), so runs can be reproduced exactly.
        """
        n = self.nets
        t = self.lengths_of(x) if t is None else t
        noise = list(noise) if noise is not None else None
        out = {"g_loss": [], "e_loss": []}
        for _ in range(10):
            z = noise.pop(0) if noise is not None else torch.rand_like(x)
            self._zero("emb", "rec", "sup", "g", "d")
            g_loss = losses.generator_forward(n["emb"], n["sup"], n["g"], n["d"], n["rec"], x, t, z,
                                              group=self.group)
            g_loss.backward()
            out["g_loss"].append(float(g_loss.detach()))
            self._sync_grads("g", "sup")
            self.opt["g"].step()
            self.opt["sup"].step()

            self._zero("emb", "rec", "sup")
            e_loss, _, _ = losses.embedding_forward_main(n["emb"], n["sup"], n["rec"], x, t, group=self.group)
            e_loss.backward()
            out["e_loss"].append(float(e_loss.detach()))
            self._sync_grads("emb", "rec")
            self.opt["emb_main"].step()
            self.opt["rec"].step()

        z = noise.pop(0) if noise is not None else torch.rand_like(x)
        self._zero("emb", "sup", "g", "d")
        d_loss = losses.discriminator_forward(n["emb"], n["sup"], n["g"], n["d"], x, t, z, group=self.group)
        # the loss value is global (identical on every rank), so all ranks take the same branch
        stepped = bool(d_loss > self.d_threshold)
        if stepped:
            d_loss.backward()
            self._sync_grads("d")
            self.opt["d"].step()
        out["d_loss"] = float(d_loss.detach())
        out["d_stepped"] = stepped
        return out

    @torch.no_grad()
    def generate(self, z: torch.Tensor, t=None) -> torch.Tensor:
        n = self.nets
        t = [z.shape[1]] * z.shape[0] if t is None else t
        return losses.inference(n["sup"], n["g"], n["rec"], z, t)

    def state_dicts(self) -> Dict[str, Dict[str, torch.Tensor]]:
        return {k: m.state_dict() for k, m in self.nets.items()}

    def load_state_dicts(self, sds: Dict[str, Dict[str, torch.Tensor]]) -> None:
        for k, sd in sds.items():
            self.nets[k].load_state_dict(sd)
