"""Test infrastructure: dumps the checkpoint key maps of the reference's converters
(tools/model_conversion.py:8-242 sdwebui UNet, :272-509 diffusers UNet, :511-685 diffusers VAE) by
importing them from /root/reference, so tests/test_host.py can pin lib/weights_io.py's derived maps.
Run in the build container only:  python oracle/make_keymap_golden.py"""
import importlib.util
import json
import os

REF = "/root/reference/tools/model_conversion.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "keymaps.json")


def main():
    spec = importlib.util.spec_from_file_location("ref_model_conversion", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    out = {}
    out["sdwebui_unet"] = m.sdwebui_diffuser_to_pfd_mover().get_mapping()
    for name in ("sdhuggingface_diffuser_to_pfd_mover", "sdhuggingface_vae_to_pfd_mover"):
        obj = getattr(m, name)()
        if hasattr(obj, "get_mapping"):
            out[name] = obj.get_mapping()
    import torch

    def tag(entry):  # [from, to] or [from, to, fn]: name the tensor transform by what it does to a probe
        if len(entry) == 2:
            return list(entry)
        shape = tuple(entry[2](torch.zeros(2, 3)).shape)
        assert shape == (2, 3, 1, 1), shape
        return [entry[0], entry[1], "unsqueeze_hw"]

    out = {k: [tag(e) for e in v] for k, v in out.items()}
    json.dump(out, open(OUT, "w"), indent=0)
    print({k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
