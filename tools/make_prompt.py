#!/usr/bin/env python3
"""Write the prompt-embedding file inference.py consumes (--prompt_embeds): {'caption_embeds': [1, L, 4096], 'emb_mask': [1, L]}.

The reference produces such files with its training-side helper (test_scripts/test_controlnet.py:383-395: T5Tokenizer(max_length,
padding="max_length", truncation) -> T5EncoderModel -> torch.save) and ships one for the fixed restoration prompt; this tool is that
producer on the MI355X path: tokenizer from a LOCAL folder (spiece.model + tokenizer_config.json, as DeepFloyd/t5-v1_1-xxl or the
`tokenizer/` subfolder of a PixArt pipeline lays them out), text encoder = instarevive_amd.models.T5EncoderModel (HIP) loaded from the
matching weight folder.

    python tools/make_prompt.py --t5 /models/t5-v1_1-xxl --prompt "a high quality photo" --max_length 300 --out prompt_embeds.pth
    python tools/make_prompt.py --pipeline /models/PixArt-XL-2-512x512 --prompt "" --out null_embed.pth     (tokenizer/, text_encoder/ subfolders)

--clean applies the reference's caption cleaning first (T5Embedder.text_preprocessing, diffusion/model/t5.py:106-233; the helper above
feeds the raw prompt, which is the default here too)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_prompt(tokenizer, text_encoder, prompt: str, max_length: int, clean: bool = False):
    """-> {'caption_embeds': fp32 [1, L, d_model] (cpu), 'emb_mask': int64 [1, L]} exactly as test_controlnet.py:389-395 saves them."""
    import torch
    if clean:
        from instarevive_amd.captions import text_preprocessing
        prompt = text_preprocessing(prompt)
    tok = tokenizer(prompt, max_length=max_length, padding="max_length", truncation=True, return_tensors="pt")
    ids, mask = tok["input_ids"], tok["attention_mask"]
    dev = text_encoder.device
    emb = text_encoder(ids.to(dev), attention_mask=mask.to(dev))["last_hidden_state"]   # (the reference indexes the HF output with [0]: the same tensor)
    return {"caption_embeds": emb.detach().to("cpu", torch.float32), "emb_mask": mask.to("cpu")}


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--t5", help="folder with the T5 tokenizer files and encoder weights (DeepFloyd/t5-v1_1-xxl layout)")
    ap.add_argument("--pipeline", help="diffusers pipeline folder with tokenizer/ and text_encoder/ subfolders (PixArt-XL-2-*)")
    ap.add_argument("--prompt", required=True)
    ap.add_argument("--max_length", type=int, default=300)
    ap.add_argument("--clean", action="store_true", help="apply T5Embedder.text_preprocessing (clean_caption twice) to the prompt first")
    ap.add_argument("--out", required=True)
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args()
    if bool(args.t5) == bool(args.pipeline):
        raise SystemExit("give exactly one of --t5 / --pipeline")
    import torch
    from transformers import T5Tokenizer
    from instarevive_amd.models import T5EncoderModel
    if args.pipeline:
        tokenizer = T5Tokenizer.from_pretrained(args.pipeline, subfolder="tokenizer")
        enc = T5EncoderModel.from_pretrained(args.pipeline, subfolder="text_encoder")
    else:
        tokenizer = T5Tokenizer.from_pretrained(args.t5)
        enc = T5EncoderModel.from_pretrained(args.t5)
    enc.to(torch.device(args.device))
    out = make_prompt(tokenizer, enc, args.prompt, args.max_length, args.clean)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)) or ".", exist_ok=True)
    torch.save(out, args.out)
    print(f"saved {args.out}: caption_embeds {tuple(out['caption_embeds'].shape)}, {int(out['emb_mask'].sum())} of {out['emb_mask'].shape[-1]} tokens")


if __name__ == "__main__":
    main()
