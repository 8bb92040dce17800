"""text_cache.BertFeatureCache: cached per-description encoder features == the reference's per-batch
tokenizer(padding=True) + BertModel call (diffusion_scene_layout_ddpm.py:216-219), on a small random-init BERT (the real
weights cannot be downloaded here; the identity being tested does not depend on them)."""
import torch


def _tiny_bert(tmp_path):
    from transformers import BertConfig, BertModel, BertTokenizer
    words = "[PAD] [UNK] [CLS] [SEP] [MASK] the room has a bed two nightstands and wardrobe there is desk chair next to".split()
    vocab = tmp_path / "vocab.txt"
    vocab.write_text("\n".join(words) + "\n")
    tok = BertTokenizer(str(vocab), do_lower_case=True)
    torch.manual_seed(0)
    model = BertModel(BertConfig(vocab_size=len(words), hidden_size=32, num_hidden_layers=2, num_attention_heads=4,
                                 intermediate_size=64, max_position_embeddings=40)).eval()
    return tok, model


def test_cached_features_equal_the_per_batch_encoder_call(tmp_path):
    from diffuscene_amd.text_cache import BertFeatureCache
    tok, model = _tiny_bert(tmp_path)
    texts = ["the room has a bed", "there is a desk and a chair next to the bed", "two nightstands", "the room has a bed"]
    cache = BertFeatureCache(tok, model, max_tokens=24)
    got = cache.batch(texts, "cpu")
    with torch.no_grad():
        ref = model(**tok(texts, return_tensors="pt", padding=True)).last_hidden_state
    assert got.shape == ref.shape and len(cache) == 3                     # duplicates are encoded once
    assert float((got - ref).abs().max()) < 2e-6 * float(ref.abs().max() + 1)
    # a different batch composition pads to a different length: still the direct call
    sub = [texts[2], texts[0]]
    with torch.no_grad():
        ref2 = model(**tok(sub, return_tensors="pt", padding=True)).last_hidden_state
    got2 = cache.batch(sub, "cpu")
    assert got2.shape == ref2.shape and float((got2 - ref2).abs().max()) < 2e-6 * float(ref2.abs().max() + 1)
    # round trip through state_dict
    c2 = BertFeatureCache(tok, model, max_tokens=24).load_state_dict(cache.state_dict())
    assert torch.equal(c2.batch(sub, "cpu"), got2)
    sp = cache.attach_to_samples({"description": sub, "class_labels": torch.zeros(2, 3, 4)})
    assert torch.equal(sp["desc_bert"], got2)


def test_truncation_is_an_error_not_a_silent_difference(tmp_path):
    import pytest
    from diffuscene_amd.text_cache import BertFeatureCache
    tok, model = _tiny_bert(tmp_path)
    cache = BertFeatureCache(tok, model, max_tokens=6)
    cache.batch(["two nightstands"], "cpu")                                # 4 tokens with [CLS] / [SEP]: fits
    with pytest.raises(ValueError, match="does not truncate"):
        cache.batch(["there is a desk and a chair next to the bed"], "cpu")
