"""Drop-in counterpart of the reference's inference_recon.py (same flags), running on g2vlm_amd.
Fixes the reference's Namespace-as-path bug and sorts the folder listing."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from g2vlm_utils import load_model_and_tokenizer, save_ply_visualization  # noqa: E402  (the reference's import line, inference_recon.py:15)

parser = argparse.ArgumentParser(description="Demo for 3D visualization")
parser.add_argument("--image_folder", type=str, default="examples/dl3dv/", help="Path to folder containing images")
parser.add_argument("--model_path", type=str, default="InternRobotics/G2VLM-2B-MoT")
parser.add_argument("--save_path", type=str, default="results/arkitscenes_results.ply")


def main(argv=None):
    args = parser.parse_args(argv)
    names = sorted(n for n in os.listdir(args.image_folder) if n.lower().endswith((".png", ".jpg", ".jpeg")))
    image_names = [os.path.join(args.image_folder, n) for n in names]
    print(image_names)
    model, tokenizer, new_token_ids, vit_image_transform, dino_transform = load_model_and_tokenizer(args.model_path)
    pred = model.recon(tokenizer, new_token_ids, dino_transform, image_names)
    save_ply_visualization(pred, args.save_path)
    return pred


if __name__ == "__main__":
    main()
