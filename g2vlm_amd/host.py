"""Host-side logic of the hot path: image loading / preprocessing and the packed-sequence
bookkeeping of the reference's `prepare_*` methods, in closed form.

Everything here is CPU integer / PIL work ("host"); no model math.
"""
import math

import numpy as np
import torch

RESNET_MEAN = [0.485, 0.456, 0.406]
RESNET_STD = [0.229, 0.224, 0.225]
OPENAI_CLIP_MEAN = [0.48145466, 0.4578275, 0.40821073]
OPENAI_CLIP_STD = [0.26862954, 0.26130258, 0.27577711]


# ----------------------------------------------------------------------------- image loading
def load_images(images, new_width):
    """reference data/transforms_vggt.py:411-451: PIL open, LANCZOS resize of EVERY image to the
    target size computed from the FIRST one, ToTensor.  Unlike the reference, a failing image
    raises instead of being silently dropped (SURVEY a1 note)."""
    from PIL import Image
    if isinstance(images[0], str):
        images = [Image.open(p) for p in images]
    w0, h0 = images[0].size
    if new_width is None:
        tw, th = max(1, round(w0 / 14)) * 14, max(1, round(h0 / 14)) * 14
    else:
        tw, th = new_width, round(h0 * (new_width / w0) / 14) * 14
    out = []
    for im in images:
        r = im.resize((tw, th), Image.Resampling.LANCZOS)
        a = np.asarray(r.convert("RGB") if r.mode != "RGB" else r, dtype=np.uint8)
        out.append(torch.from_numpy(a.copy()).permute(2, 0, 1).float().div(255))
    return torch.stack(out, 0)


def load_images_u8(images, new_width, device=None):
    """load_images up to, not including, ToTensor: the resized frames as uint8 [N,H,W,3] (ToTensor is exactly k/255, which
    the device preprocessing kernel reproduces bit for bit, so only a quarter of the bytes cross PCIe).
    device: the LANCZOS resize itself runs there (hip.lanczos_resize_u8, bit-exact with Pillow's): the decoded RGB frames are
    uploaded as they are and the result is a device tensor.  Images Pillow would not resample as plain 8-bit RGB (palette,
    alpha, greyscale, 16-bit) take the host path for the whole call."""
    from PIL import Image
    if isinstance(images[0], str):
        images = [Image.open(p) for p in images]
    w0, h0 = images[0].size
    if new_width is None:
        tw, th = max(1, round(w0 / 14)) * 14, max(1, round(h0 / 14)) * 14
    else:
        tw, th = new_width, round(h0 * (new_width / w0) / 14) * 14
    if device is not None and torch.device(device).type == "cuda" and all(im.mode == "RGB" for im in images):
        from . import hip
        out = torch.empty((len(images), th, tw, 3), dtype=torch.uint8, device=device)
        by_size = {}
        for i, im in enumerate(images):
            by_size.setdefault(im.size, []).append(i)
        for (w, h), idx in by_size.items():                  # one upload + one resize per source size
            raw = np.empty((len(idx), h, w, 3), dtype=np.uint8)
            for j, i in enumerate(idx):
                raw[j] = np.asarray(images[i], dtype=np.uint8)
            res = hip.lanczos_resize_u8(hip.h2d(torch.from_numpy(raw), device), th, tw)
            if len(by_size) == 1:
                return res
            out[torch.tensor(idx, device=device)] = res
        return out
    out = np.empty((len(images), th, tw, 3), dtype=np.uint8)
    for i, im in enumerate(images):
        r = im.resize((tw, th), Image.Resampling.LANCZOS)
        out[i] = np.asarray(r.convert("RGB") if r.mode != "RGB" else r, dtype=np.uint8)
    return torch.from_numpy(out)


# ---- Pillow's LANCZOS resample, restated so that it can run on the device --------------------------------------------------
# The reference resizes with `img.resize((W, H), Image.Resampling.LANCZOS)` (data/transforms_vggt.py:437); Pillow is a
# third-party dependency of it (not under /root/reference; importable here), so parity is pinned on Pillow's own output.
# Pillow (src/libImaging/Resample.c, 8 bits per channel): per output coordinate a window of ceil(3 * max(scale, 1)) * 2 + 1
# taps of sinc(x) sinc(x / 3) evaluated in double and normalised to sum 1 (precompute_coeffs), rounded to 22-bit fixed point
# (normalize_coeffs_8bpc); a horizontal pass over the source rows the vertical pass will need, then a vertical pass, each
# `clip8((2^21 + sum(pixel * k)) >> 22)` on uint8 data.  Integer arithmetic: the device kernel is bit-exact by construction
# once the tables agree, and the tables are built here on the host (math.sin = the libm call Pillow's C code makes).
PIL_PRECISION_BITS = 32 - 8 - 2
_LANCZOS_TABLES = {}


def _lanczos(x):
    if not (-3.0 <= x < 3.0):
        return 0.0

    def sinc(v):
        if v == 0.0:
            return 1.0
        v = v * math.pi
        return math.sin(v) / v
    return sinc(x) * sinc(x / 3.0)


def lanczos_tables(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc for the full-image box: (bounds int32 [out, 2] = (first tap, taps),
    coefficients int32 [out, ksize])."""
    key = (int(in_size), int(out_size))
    if key not in _LANCZOS_TABLES:
        scale = float(np.float32(in_size) - np.float32(0.0)) / out_size          # box edges are C floats in Pillow
        filterscale = max(scale, 1.0)
        support = 3.0 * filterscale
        ksize = int(math.ceil(support)) * 2 + 1
        bounds = np.zeros((out_size, 2), dtype=np.int32)
        kk = np.zeros((out_size, ksize), dtype=np.int32)
        ss = 1.0 / filterscale
        for xx in range(out_size):
            center = 0.0 + (xx + 0.5) * scale
            xmin = max(int(center - support + 0.5), 0)
            xmax = min(int(center + support + 0.5), in_size) - xmin
            w = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
            ww = 0.0
            for v in w:
                ww += v
            for x in range(xmax):
                k = w[x] / ww if ww != 0.0 else w[x]
                kk[xx, x] = int(-0.5 + k * (1 << PIL_PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PIL_PRECISION_BITS))
            bounds[xx] = (xmin, xmax)
        _LANCZOS_TABLES[key] = (torch.from_numpy(bounds), torch.from_numpy(kk))
    return _LANCZOS_TABLES[key]


def lanczos_resize_u8_reference(frames, out_h, out_w):
    """The two passes on the host in numpy int64 (tests: pins the tables and the pass order against Pillow without a GPU).
    frames uint8 [N, H, W, 3] -> uint8 [N, out_h, out_w, 3]."""
    a = frames.numpy() if torch.is_tensor(frames) else np.asarray(frames)
    n, h, w, _ = a.shape
    half = 1 << (PIL_PRECISION_BITS - 1)
    if w != out_w:
        bnd, kk = (t.numpy() for t in lanczos_tables(w, out_w))
        out = np.empty((n, h, out_w, 3), dtype=np.uint8)
        for xx in range(out_w):
            x0, cnt = bnd[xx]
            acc = half + (a[:, :, x0:x0 + cnt, :].astype(np.int64) * kk[xx, :cnt].astype(np.int64)[None, None, :, None]).sum(2)
            out[:, :, xx] = np.clip(acc >> PIL_PRECISION_BITS, 0, 255)
        a = out
    if h != out_h:
        bnd, kk = (t.numpy() for t in lanczos_tables(h, out_h))
        out = np.empty((n, out_h, a.shape[2], 3), dtype=np.uint8)
        for yy in range(out_h):
            y0, cnt = bnd[yy]
            acc = half + (a[:, y0:y0 + cnt].astype(np.int64) * kk[yy, :cnt].astype(np.int64)[None, :, None, None]).sum(1)
            out[:, yy] = np.clip(acc >> PIL_PRECISION_BITS, 0, 255)
        a = out
    return torch.from_numpy(np.ascontiguousarray(a))


def load_and_resize14(images, new_width=518, patch=14):
    """reference data/transforms_vggt.py:454-462 (the trailing bilinear resize is identity-sized for width 518)."""
    if torch.is_tensor(images):
        imgs = images
    else:
        imgs = load_images(images, new_width)
    h, w = imgs.shape[-2:]
    ph, pw = h // patch, w // patch
    if (ph * patch, pw * patch) != (h, w):
        imgs = torch.nn.functional.interpolate(imgs, (ph * patch, pw * patch), mode="bilinear", align_corners=False, antialias=True)
    return imgs


def load_and_resize16(images, new_width=518):
    """reference data/transforms_vggt.py:464-471: the same loader (its /14 height rounding included), then an antialiased
    bilinear resize down to multiples of 16 (518 -> 512 wide) - the loader of the use_dinov3 variant."""
    return load_and_resize14(images, new_width, patch=16)


def smart_resize(height, width, factor=28, min_pixels=56 * 56, max_pixels=14 * 14 * 4 * 1280):
    """reference modeling/qwen2vl/image_processing_qwen2_vl.py:56-84"""
    if height < factor or width < factor:
        raise ValueError(f"height:{height} or width:{width} must be larger than factor:{factor}")
    if max(height, width) / min(height, width) > 200:
        raise ValueError(f"absolute aspect ratio must be smaller than 200, got {max(height, width) / min(height, width)}")
    h_bar = round(height / factor) * factor
    w_bar = round(width / factor) * factor
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = math.floor(height / beta / factor) * factor
        w_bar = math.floor(width / beta / factor) * factor
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar = math.ceil(height * beta / factor) * factor
        w_bar = math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


class QwenVL2ImageTransform:
    """reference data/transforms.py:151-178 + Qwen2VLImageProcessor._preprocess
    (image_processing_qwen2_vl.py:155-273), constructed from defaults — no hub fetch by name.
    __call__(list[PIL]) -> (pixel_values [T,1176] fp32, image_grid_thw [1,3])."""

    def __init__(self, image_size_h, image_size_w, image_stride=14, min_pixels=56 * 56, max_pixels=28 * 28 * 1280, device=None,
                 k_pad=1216):
        """device: if given, everything after the PIL resize (rescale, normalise, temporal pairing, patch reorder, bf16
        cast, zero-pad of K to `k_pad` = the patch GEMM's padded K) runs in one HIP kernel on the uint8 frame and
        __call__ returns a device bf16 [T, k_pad] matrix that G2VLM.forward_cache_update_vit consumes as is - a
        quarter of the bytes over PCIe and no host transposes."""
        self.img_h, self.img_w, self.patch = image_size_h, image_size_w, image_stride
        self.min_pixels, self.max_pixels = min_pixels, max_pixels
        self.merge, self.temporal = 2, 2
        self.device, self.k_pad = device, k_pad

    def __call__(self, img, img_num=1):
        from PIL import Image
        frames = []
        u8 = []
        rh = rw = None
        for ii in img:
            ii = ii.resize((self.img_w, self.img_h), 3)                               # 3 = PIL BICUBIC (reference passes 3)
            ii = ii.convert("RGB")
            w, h = ii.size
            rh, rw = smart_resize(h, w, factor=self.patch * self.merge, min_pixels=self.min_pixels, max_pixels=self.max_pixels)
            ii = ii.resize((rw, rh), Image.Resampling.BICUBIC)
            if self.device is not None:
                u8.append(np.asarray(ii, dtype=np.uint8))
                continue
            a = np.asarray(ii, dtype=np.uint8).astype(np.float64) * (1 / 255)          # HF rescale in float64 -> float32
            a = a.astype(np.float32)
            a = (a - np.array(OPENAI_CLIP_MEAN, dtype=np.float32)) / np.array(OPENAI_CLIP_STD, dtype=np.float32)
            frames.append(np.transpose(a, (2, 0, 1)))
        if self.device is not None:
            from . import hip
            fr = hip.h2d(torch.from_numpy(np.ascontiguousarray(np.stack(u8, 0))), self.device)
            gt = (len(u8) + 1) // 2
            return hip.qwen_patchify_u8(fr, OPENAI_CLIP_MEAN, OPENAI_CLIP_STD, self.k_pad), torch.tensor([[gt, rh // self.patch, rw // self.patch]])
        p = np.stack(frames, 0)
        if p.shape[0] % self.temporal != 0:
            p = np.concatenate([p, np.repeat(p[-1][None], self.temporal - 1, 0)], 0)
        ch = p.shape[1]
        gt, gh, gw = p.shape[0] // self.temporal, rh // self.patch, rw // self.patch
        p = p.reshape(gt, self.temporal, ch, gh // self.merge, self.merge, self.patch, gw // self.merge, self.merge, self.patch)
        p = p.transpose(0, 3, 6, 4, 7, 2, 1, 5, 8)
        flat = p.reshape(gt * gh * gw, ch * self.temporal * self.patch * self.patch)
        return torch.from_numpy(np.ascontiguousarray(flat)), torch.tensor([[gt, gh, gw]])


class DinoImageNormalizeTransform:
    """reference data/transforms_vggt.py:27-45 (unused downstream, kept for the 5-tuple contract)."""

    def __init__(self, mode="crop", target_size=518):
        self.mode, self.target_size = mode, target_size

    def __call__(self, img, img_num=1):
        m = torch.tensor(RESNET_MEAN).view(-1, 1, 1); s = torch.tensor(RESNET_STD).view(-1, 1, 1)
        return (img - m) / s


# ----------------------------------------------------------------------------- bookkeeping
def prepare_text(curr_kvlens, curr_rope, prompts, tokenizer, new_token_ids, bos=False, eos=False, eos_bos_assistant=False):
    """reference g2vlm.py:561-699 (the four prepare_prompts* variants differ only in the framing)."""
    ids_all, pos_all, lens, idx, kv_idx = [], [], [], [], []
    curr = 0
    newlens, new_rope = [], []
    for prompt, kvlen, pos in zip(prompts, curr_kvlens, curr_rope):
        kv_idx.extend(range(curr, curr + kvlen)); curr += kvlen
        ids = list(tokenizer.encode(prompt))
        if eos_bos_assistant:
            ids = ids + [new_token_ids["eos_token_id"], new_token_ids["bos_token_id"]] + list(tokenizer.encode("assistant\n"))
        if bos:
            ids = [new_token_ids["bos_token_id"]] + ids
        if eos:
            ids = ids + [new_token_ids["eos_token_id"]]
        lens.append(len(ids)); ids_all.extend(ids)
        pos_all.extend(range(pos, pos + len(ids)))
        idx.extend(range(curr, curr + len(ids)))
        newlens.append(kvlen + len(ids)); new_rope.append(pos + len(ids))
        curr += len(ids)
    gi = {
        "text_token_lens": torch.tensor(lens, dtype=torch.int),
        "packed_text_ids": torch.tensor(ids_all, dtype=torch.long),
        "packed_text_position_ids": torch.tensor(pos_all, dtype=torch.long).expand(3, -1),
        "packed_text_indexes": torch.tensor(idx, dtype=torch.long),
        "packed_key_value_indexes": torch.tensor(kv_idx, dtype=torch.long),
        "key_values_lens": torch.tensor(list(curr_kvlens), dtype=torch.int),
    }
    return gi, newlens, new_rope


def prepare_image_tokens(curr_kvlen, curr_pos, grids, new_token_ids, merge=1):
    """Closed form of the per-image loops of prepare_dino_images_pi3 (g2vlm.py:868-966, merge=1) and
    prepare_vit_images (:735-810, merge=2) with get_rope_index_image_3D[_dino] (data_utils.py:78-201):
    per image [<|vision_start|>, t*h*w grid tokens, <|vision_end|>]; grid token (r,c) sits at
    (base, base+r, base+c); the next position is base + (max-min over the grid) + 1."""
    text_ids, text_idx, tok_idx, pos_parts, tok_lens = [], [], [], [], []
    local = 0
    pos = curr_pos
    for (t, h, w) in grids:
        h, w = h // merge, w // merge
        n = t * h * w
        text_ids += [new_token_ids["start_of_image"], new_token_ids["end_of_image"]]
        text_idx += [local, local + n + 1]
        tok_idx.append(torch.arange(local + 1, local + 1 + n))
        pos_parts.append(torch.full((3, 1), pos, dtype=torch.long)); pos += 1
        ti = torch.arange(t).view(-1, 1, 1).expand(t, h, w).reshape(-1)
        hi = torch.arange(h).view(1, -1, 1).expand(t, h, w).reshape(-1)
        wi = torch.arange(w).view(1, 1, -1).expand(t, h, w).reshape(-1)
        pos_parts.append(torch.stack([ti, hi, wi], 0) + pos)
        pos += max(t, h, w) - 1 + 1
        pos_parts.append(torch.full((3, 1), pos, dtype=torch.long)); pos += 1
        tok_lens.append(n)
        local += n + 2
    gi = {
        "packed_text_ids": torch.tensor(text_ids, dtype=torch.long),
        "packed_text_indexes": torch.tensor(text_idx, dtype=torch.long),
        "token_seqlens": torch.tensor(tok_lens, dtype=torch.int),
        "packed_token_indexes": torch.cat(tok_idx) if tok_idx else torch.zeros(0, dtype=torch.long),
        "packed_position_ids": torch.cat(pos_parts, dim=1),
        "packed_seqlens": torch.tensor([local], dtype=torch.int),
        "packed_indexes": torch.arange(curr_kvlen, curr_kvlen + local),
        "packed_key_value_indexes": torch.arange(curr_kvlen),
        "key_values_lens": torch.tensor([curr_kvlen], dtype=torch.int),
    }
    return gi, curr_kvlen + local, pos


def vit_rot_pos(t, h, w, head_dim, merge=2):
    """rot_pos_emb + VisionRotaryEmbedding (reference modeling_qwen2_vl.py:249-258, 1019-1052):
    returns cos, sin fp32 [t*h*w, head_dim] in merge-block token order."""
    hp = torch.arange(h).unsqueeze(1).expand(-1, w).reshape(h // merge, merge, w // merge, merge).permute(0, 2, 1, 3).flatten()
    wp = torch.arange(w).unsqueeze(0).expand(h, -1).reshape(h // merge, merge, w // merge, merge).permute(0, 2, 1, 3).flatten()
    pos_ids = torch.stack([hp, wp], dim=-1).repeat(t, 1)
    dim = head_dim // 2
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float) / dim))
    freqs = torch.outer(torch.arange(max(h, w), dtype=torch.float), inv_freq)
    rot = freqs[pos_ids].flatten(1)
    emb = torch.cat((rot, rot), dim=-1)
    return emb.cos(), emb.sin()


# ----------------------------------------------------------------------------- PLY output
def write_ply_binary(path, points, colors01):
    """Binary little-endian PLY with float xyz + uchar rgb (open3d-free stand-in for
    o3d.io.write_point_cloud in reference g2vlm_utils.py:146-149)."""
    pts = np.asarray(points, dtype="<f4").reshape(-1, 3)
    col = (np.clip(np.asarray(colors01, dtype=np.float64).reshape(-1, 3), 0, 1) * 255).astype(np.uint8)
    rec = np.empty(len(pts), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1")])
    rec["x"], rec["y"], rec["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
    rec["r"], rec["g"], rec["b"] = col[:, 0], col[:, 1], col[:, 2]
    hdr = ("ply\nformat binary_little_endian 1.0\n" f"element vertex {len(pts)}\n"
           "property float x\nproperty float y\nproperty float z\n"
           "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n")
    with open(path, "wb") as f:
        f.write(hdr.encode("ascii"))
        f.write(rec.tobytes())
