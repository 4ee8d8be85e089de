#!/usr/bin/env python3
"""Command-line entry, same flags as the reference's ``main.py:11-109``:

    python main.py -p infer -m mpp -c mpp_hrcM [-d DATASET] [-o]

``-m mpp`` runs the MI355X sampler; ``-m posnet`` / ``-m shapenet`` with ``-p infer`` write the score-map
hand-off pickles the reference's MPP stage reads (``NNNN_results.pkl``).  ``-p train -m mpp`` learns the
energy weights (manual / ordering / integral criterion) and calibrates; the training loops of the two
U-Nets are outside this build.  With ``torchrun --nproc-per-node N`` the images of the dataset are
dealt to N GPUs (one gather of the results at the end; RCCL).
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("-m", "--model", help="model to use")
    parser.add_argument("-d", "--dataset", help="dataset to use, defaults to the one specified in config")
    parser.add_argument("-p", "--procedure", help="procedure to execute")
    parser.add_argument("-c", "--config", help="model config file, or the name of a stored model")
    parser.add_argument("-o", "--overwrite", action="store_true", help="overwrite existing results")
    parser.add_argument("-r", "--resume", action="store_true", help="(training) resume from checkpoint")
    parser.add_argument("--spec-waves", type=int, default=None,
                        help="speculative waves per chain (1,2,4,8,16); default: chosen per launch from the number of tiles and the "
                             "LDS footprint of a chain (sampler.choose_spec_waves)")
    parser.add_argument("--unet", action="store_true", help="compute the score maps with the U-Nets on the GPU "
                                                            "instead of reading NNNN_results.pkl")
    args = parser.parse_args()

    from mpp_cnn_rs_object_detection_amd.paths import get_model_base_path, resolve_model_config_path
    with open(resolve_model_config_path(args.config)) as f:
        config = json.load(f)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    overwrite_results = args.overwrite and args.procedure != "train"

    if args.model == "mpp":
        from mpp_cnn_rs_object_detection_amd.mpp_model import MPPModel
        nets = None
        if args.unet:
            nets = load_nets(config, local_rank)
        model = MPPModel(config, phase="train" if args.procedure == "train" else "val",
                         load=args.procedure not in ["train", "data_preview"], dataset=args.dataset, device=local_rank,
                         nets=nets, spec_waves=args.spec_waves)
    elif args.model in ("posnet", "shapenet"):
        model = ScoreMapWriter(config, args.model, args.dataset, local_rank)
    else:
        raise ValueError(f"model {args.model!r}: only mpp / posnet / shapenet inference is built")

    if args.procedure == "infer":
        print("infering on dataset")
        model.infer(subset="val", min_confidence=0.2, display_min_confidence=0.5, overwrite=overwrite_results)
    elif args.procedure == "infereval":
        model.infer(subset="val", min_confidence=0.2, display_min_confidence=0.5, overwrite=overwrite_results)
        model.eval()
    elif args.procedure == "eval":
        model.eval()
    elif args.procedure == "train":
        model.train()
    else:
        raise ValueError(args.procedure)
    print("done !")


def load_nets(mpp_config, device):
    """PosNet + ShapeNet named by the MPP config, weights from <model_path>/{posnet,shapenet}/<name>/model.pt."""
    from mpp_cnn_rs_object_detection_amd import unet
    from mpp_cnn_rs_object_detection_amd.paths import get_model_base_path
    base = get_model_base_path()
    pos_dir = os.path.join(base, "posnet", mpp_config["dataset"]["position_model"])
    shp_dir = os.path.join(base, "shapenet", mpp_config["dataset"]["shape_model"])
    pos, shp = unet.PosNet(), unet.ShapeNet()
    if not unet.load_torch_model(pos, pos_dir) or not unet.load_torch_model(shp, shp_dir):
        raise FileNotFoundError(f"no model.pt / checkpoint_*.pt under {pos_dir} or {shp_dir}")
    return unet.ScoreMapNets(pos, shp, device=device, div_clf=unet.load_div_clf(pos_dir))


class ScoreMapWriter:
    """``-m posnet|shapenet -p infer``: write the reference's hand-off pickles
    (``pos_net_model.py:407-424`` / ``shape_net_model.py:353-381``)."""

    def __init__(self, config, kind, dataset, device):
        from mpp_cnn_rs_object_detection_amd import unet
        from mpp_cnn_rs_object_detection_amd.paths import get_model_base_path
        self.kind, self.config = kind, config
        self.dataset = dataset or config["data_loader"]["dataset"]
        d = os.path.join(get_model_base_path(), kind, config["model_name"])
        self.pos, self.shp = unet.PosNet(), unet.ShapeNet()
        if not unet.load_torch_model(self.pos if kind == "posnet" else self.shp, d):
            raise FileNotFoundError(f"no model.pt / checkpoint_*.pt under {d}")
        self.nets = unet.ScoreMapNets(self.pos, self.shp, device=device, div_clf=unet.load_div_clf(d))

    def infer(self, subset, overwrite=True, **_):
        import pickle
        import re
        from matplotlib import pyplot as plt
        from mpp_cnn_rs_object_detection_amd import mappings
        from mpp_cnn_rs_object_detection_amd.paths import fetch_data_paths, get_inference_path
        out_dir = get_inference_path(self.config["model_name"], self.dataset, subset)
        os.makedirs(out_dir, exist_ok=True)
        for pf in fetch_data_paths(self.dataset, subset)["images"]:
            pid = int(re.match(r"([0-9]+).*.png", os.path.split(pf)[1]).group(1))
            out = os.path.join(out_dir, f"{pid:04}_results.pkl")
            if os.path.exists(out) and not overwrite:
                continue
            det, marks = self.nets.infer(plt.imread(pf)[:, :, :3])
            if self.kind == "posnet":
                res = {"detection_map": det.cpu().numpy(), "detection_type": "map"}
            else:
                res = {"output": [m.permute(2, 0, 1).unsqueeze(0).cpu().numpy() for m in marks],
                       "mappings": mappings.default_mappings()}
            with open(out, "wb") as f:
                pickle.dump(res, f)

    def eval(self):
        raise NotImplementedError


if __name__ == "__main__":
    main()
