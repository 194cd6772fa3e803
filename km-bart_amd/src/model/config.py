"""MultiModalBartConfig with the reference's attribute surface (reference src/model/config.py:4-92).

The reference subclasses transformers.BartConfig; this build does not depend on transformers, so the
handful of PretrainedConfig behaviours the callers use (`from_dict`, `from_pretrained`,
`save_pretrained`, `to_dict`, generation defaults) are restated here.
"""
import copy
import json
import os

CONFIG_NAME = "config.json"

_DEFAULTS = dict(
    # reference src/model/config.py:4-47
    activation_dropout=0.0, extra_pos_embeddings=2, activation_function="gelu", vocab_size=50320,
    image_feature_size=2048 + 4, d_model=1024, encoder_ffn_dim=4096, encoder_layers=12,
    encoder_attention_heads=16, decoder_ffn_dim=4096, decoder_layers=12, decoder_attention_heads=16,
    encoder_layerdrop=0.0, decoder_layerdrop=0.0, attention_dropout=0.0, dropout=0.1,
    max_position_embeddings=1024, init_std=0.02, classif_dropout=0.0, num_labels=1, num_attributes=1,
    num_relations=1, is_encoder_decoder=True, pad_token_id=1, bos_token_id=0, eos_token_id=2,
    img_feat_id=50273, cls_token_id=50276, normalize_before=False, add_final_layer_norm=False,
    scale_embedding=False, normalize_embedding=True, static_position_embeddings=False, add_bias_logits=False,
    decoder_start_token_id=0, partial_load=(), lm_loss_factor=1.0, mrm_loss_factor=1.0,
    attribute_loss_factor=1.0, relation_loss_factor=1.0,
    # transformers 3.0.2 PretrainedConfig defaults the generation path reads (mixins.py:150-173)
    max_length=20, min_length=0, do_sample=False, early_stopping=False, num_beams=1, temperature=1.0, top_k=50,
    top_p=1.0, repetition_penalty=1.0, length_penalty=1.0, no_repeat_ngram_size=0, bad_words_ids=None,
    num_return_sequences=1, use_cache=True, output_attentions=False, output_hidden_states=False,
    model_type="bart",
)


class MultiModalBartConfig:
    model_type = "bart"

    def __init__(self, **kwargs):
        d = copy.deepcopy(_DEFAULTS)
        d.update(kwargs)
        for k, v in d.items():
            setattr(self, k, v)

    # what this build implements on the device; anything else fails loudly at model construction
    def check_supported(self):
        problems = []
        if self.activation_function != "gelu":
            problems.append("activation_function must be 'gelu'")
        if self.normalize_before or self.add_final_layer_norm:
            problems.append("pre-LN / final layer norm (mBART) is not implemented")
        if self.static_position_embeddings:
            problems.append("sinusoidal positions are not implemented")
        if not self.normalize_embedding:
            problems.append("normalize_embedding=False is not implemented")
        if self.encoder_layerdrop != 0.0 or self.decoder_layerdrop != 0.0:
            problems.append("LayerDrop is not implemented")
        if self.add_bias_logits:
            problems.append("add_bias_logits is not implemented")
        if problems:
            raise NotImplementedError("MultiModalBartConfig: " + "; ".join(problems))

    @classmethod
    def from_dict(cls, config_dict, **kwargs):
        d = dict(config_dict)
        d.update(kwargs)
        return cls(**d)

    @classmethod
    def from_json_file(cls, path):
        with open(path, "r", encoding="utf-8") as f:
            return cls.from_dict(json.load(f))

    @classmethod
    def from_pretrained(cls, path, **kwargs):
        if os.path.isdir(path):
            path = os.path.join(path, CONFIG_NAME)
        cfg = cls.from_json_file(path)
        for k, v in kwargs.items():
            setattr(cfg, k, v)
        return cfg

    def to_dict(self):
        out = {k: copy.deepcopy(v) for k, v in self.__dict__.items()}
        if isinstance(out.get("partial_load"), tuple):
            out["partial_load"] = list(out["partial_load"])
        out["model_type"] = self.model_type
        return out

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"

    def save_pretrained(self, save_directory):
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, CONFIG_NAME), "w", encoding="utf-8") as f:
            f.write(self.to_json_string())

    def __repr__(self):
        return "MultiModalBartConfig " + self.to_json_string()
