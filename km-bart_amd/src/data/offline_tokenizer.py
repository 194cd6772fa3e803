"""Base vocabularies for ConditionTokenizer when `facebook/bart-large` cannot be downloaded.

`load_base_tokenizer(name_or_path)` resolves, in order: a `tokenizers` JSON file, a directory / cached name holding
BART's vocab.json + merges.txt (transformers.BartTokenizer, local files only), and otherwise fails loudly -- a
silently different vocabulary would change every id.  `train_byte_level_bpe` builds a small byte-level BPE (the same
pre-tokenisation and special-token order as BART: <s>=0, <pad>=1, </s>=2, <unk>=3) for synthetic runs and tests.
"""
import os

BART_SPECIALS = ["<s>", "<pad>", "</s>", "<unk>", "<mask>"]

_DEMO_CORPUS = [
    "PersonX walks into the kitchen and opens the fridge to get a cold drink",
    "a man is holding a red umbrella in the rain near the bus stop",
    "two dogs are playing with a ball on the green grass in the park",
    "PersonX wants to ask the woman behind the counter for directions to the station",
    "the tall building next to the river has many small windows and a blue door",
    "before PersonX needed to buy a ticket and after PersonX will sit down and read",
]


def wrap_tokenizers_object(tok=None, path=None):
    """tokenizers.Tokenizer (or its JSON file) -> transformers fast tokenizer with BART's special-token names"""
    from transformers import PreTrainedTokenizerFast
    kw = dict(bos_token="<s>", eos_token="</s>", pad_token="<pad>", unk_token="<unk>", mask_token="<mask>")
    if path is not None:
        return PreTrainedTokenizerFast(tokenizer_file=path, **kw)
    return PreTrainedTokenizerFast(tokenizer_object=tok, **kw)


def train_byte_level_bpe(corpus=None, vocab_size=1000, save_to=None):
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=vocab_size, special_tokens=BART_SPECIALS, show_progress=False,
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet())
    tok.train_from_iterator(list(corpus) if corpus is not None else _DEMO_CORPUS * 3, trainer)
    if save_to:
        tok.save(save_to)
    return wrap_tokenizers_object(tok)


def load_base_tokenizer(name_or_path="facebook/bart-large"):
    if os.path.isfile(name_or_path) and name_or_path.endswith(".json"):
        return wrap_tokenizers_object(path=name_or_path)
    try:
        from transformers import BartTokenizer
        return BartTokenizer.from_pretrained(name_or_path, local_files_only=True)
    except Exception as e:  # no vocabulary on disk and no network
        raise RuntimeError(
            "cannot load the BART vocabulary '%s' from local files (%s: %s). Pass a tokenizers JSON file "
            "(--tokenizer_json / base_tokenizer=load_base_tokenizer(path)) or build one with "
            "src.data.offline_tokenizer.train_byte_level_bpe()." % (name_or_path, type(e).__name__, e)) from e
