"""Drop-in counterpart of the reference's inference_chat.py (same flags), running on g2vlm_amd.
--image-path is honoured and an empty --question falls back to the built-in one (H6)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from g2vlm_utils import load_model_and_tokenizer, build_transform, process_conversation  # noqa: E402  (the reference's import line, inference_chat.py:8)

parser = argparse.ArgumentParser()
parser.add_argument("--model-path", type=str, default="InternRobotics/G2VLM-2B-MoT")
parser.add_argument("--image-path", type=str, default="examples/25_0.jpg")
parser.add_argument("--question", type=str, default="")


def main(argv=None):
    args = parser.parse_args(argv)
    from PIL import Image
    # the reference hands the whole Namespace to the loader (inference_chat.py:18); the loader accepts both forms
    model, tokenizer, new_token_ids, vit_image_transform, dino_transform = load_model_and_tokenizer(args)
    image_transform = build_transform(pixel=768)
    total_params = sum(p.numel() for p in model.parameters()) / 1e9
    print(f"[test] total_params: {total_params}B")
    question = ("If the table (red point) is positioned at 2.6 meters, estimate the depth of the clothes (blue point).  "
                "Calculate or judge based on the 3D center points of these objects. The unit is meter. "
                "Submit your response as one numeric value only.")
    templated = "\n" + question + "\n" + "Please answer the question using a single word or phrase."
    if args.question:
        templated = args.question
    images = [Image.open(args.image_path).convert("RGB")]
    images, conversation = process_conversation(images, templated)
    response = model.chat_with_recon(tokenizer, new_token_ids, image_transform, dino_transform, images=images,
                                     prompt=conversation, max_length=100)
    print("answer: ", response)
    return response


if __name__ == "__main__":
    main()
