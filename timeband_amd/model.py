"""TIMEBANDModel: builds / saves / loads the two networks (reference source/model.py:9-103).

Checkpoints keep the reference's file names (netD.pth, netG_<score>.pth, netD_BEST.pth ...) but
hold state_dicts with the reference's parameter names instead of pickled modules, so they load with
`weights_only=True` (nothing in the file is executed)."""
import os

import torch

from .utils.lstm_layer import LSTMGenerator as NetG
from .utils.lstm_layer import LSTMDiscriminator as NetD


class TIMEBANDModel:
    def __init__(self, config: dict) -> None:
        self.logger = config["logger"]
        self.set_config(config)
        self.netD = None
        self.netG = None

    def set_config(self, config: dict) -> None:
        self.__dict__ = {**config, **self.__dict__}

    def initiate(self, dims: dict) -> None:
        if self.netD and self.netG:
            return
        enc_dim, dec_dim = dims["encode"], dims["decode"]
        # D before G: the reference's construction (= RNG) order, model.py:56-57
        netD = NetD(dec_dim, self.hidden_dim, self.layers_num, self.device)
        netG = NetG(enc_dim, dec_dim, self.observed_len, self.forecast_len, self.hidden_dim, self.layers_num,
                    self.device)
        self.netD, self.netG = netD.to(self.device), netG.to(self.device)
        self.save()

    def get_path(self, target: str, postfix: str = "") -> str:
        name = target if postfix == "" else f"{target}_{postfix}"
        return os.path.join(self.models_path, f"{name}.pth")

    def _write(self, postfix):
        os.makedirs(self.models_path, exist_ok=True)
        for tag, net in (("netD", self.netD), ("netG", self.netG)):
            torch.save({k: v.detach().cpu() for k, v in net.state_dict().items()}, self.get_path(tag, postfix))

    def save(self, postfix: str = "", best: bool = False) -> None:
        if not getattr(self, "save_opt", False):
            return
        self._write(postfix)
        if best:
            self._write("BEST")
            postfix = f"Best({postfix})"
        self.logger.info(f"*** {postfix} MODEL IS SAVED ***")

    def load(self, postfix: str = ""):
        if getattr(self, "load_opt", False):
            pD, pG = self.get_path("netD", postfix), self.get_path("netG", postfix)
            if os.path.exists(pD) and os.path.exists(pG):
                self.logger.info(f" - {postfix} Model Loading : {pD}, {pG}")
                for path, net in ((pD, self.netD), (pG, self.netG)):
                    try:
                        sd = torch.load(path, map_location="cpu", weights_only=True)
                    except Exception as e:     # a whole pickled module (reference model.py:84-95): never unpickled here
                        raise RuntimeError(
                            f"{path} is not a state_dict checkpoint ({type(e).__name__}); the reference saves whole "
                            "pickled modules -- convert once where the file is trusted: "
                            "torch.save(torch.load(path, weights_only=False).state_dict(), path)") from e
                    if not isinstance(sd, dict):
                        raise RuntimeError(f"{path} is not a state_dict checkpoint")
                    net.load_state_dict(sd)      # in place: parameters keep aliasing their flat buffers
            else:
                self.logger.warn(f" - {postfix} Model Loading Fail")
                if postfix != "":
                    self.load()
        return self.netD, self.netG
