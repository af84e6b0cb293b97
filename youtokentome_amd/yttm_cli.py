"""`yttm` command line on top of the MI355X core: same commands and flags as the reference CLI
(youtokentome/yttm_cli.py:10-169: bpe | encode | decode | vocab), same stdout formats as the reference's native CLI
loops (bpe.cpp:1942-2028, utils.h:92-103).  Run as `python -m youtokentome_amd.yttm_cli ...`.

The loops themselves are C++ (youtokentome_amd/csrc/host_cli.cpp, behind yttm_encode_cli / yttm_decode_cli / yttm_vocab_cli):
stdin and stdout stay bytes end to end, invalid UTF-8 included."""

import click

from .bpe import _Core


@click.group()
def main():
    pass


@main.command()
@click.option("--data", type=click.Path(exists=True), required=True, help="Training data file path.")
@click.option("--model", type=click.Path(), required=True, help="Output model file path.")
@click.option("--vocab_size", type=click.INT, required=True, help="Number of tokens in the final vocabulary.")
@click.option("--coverage", type=click.FLOAT, default=1.0, show_default=True, help="Percentage of characters covered by the model.")
@click.option("--n_threads", type=click.INT, default=-1, show_default=True, help="Number of threads (accepted, unused: the GPU trains).")
@click.option("--pad_id", type=click.INT, default=0, show_default=True, help="Padding token id.")
@click.option("--unk_id", type=click.INT, default=1, show_default=True, help="Unknown token id.")
@click.option("--bos_id", type=click.INT, default=2, show_default=True, help="Begin of sentence token id.")
@click.option("--eos_id", type=click.INT, default=3, show_default=True, help="End of sentence token id.")
def bpe(data, model, vocab_size, coverage, n_threads, pad_id, unk_id, bos_id, eos_id):
    """Train BPE model."""
    _Core.train(data=data, model=model, vocab_size=vocab_size, coverage=coverage, n_threads=n_threads, pad_id=pad_id,
                unk_id=unk_id, bos_id=bos_id, eos_id=eos_id)


@main.command()
@click.option("--model", type=click.Path(exists=True), required=True, help="Path to file with learned model.")
@click.option("--output_type", type=click.Choice(["id", "subword"]), required=True, help="'id' or 'subword'.")
@click.option("--n_threads", type=click.INT, default=-1, show_default=True, help="Number of threads.")
@click.option("--bos", is_flag=True, help="Add tab begin of sentence.")
@click.option("--eos", is_flag=True, help="Add tab end of sentence.")
@click.option("--reverse", is_flag=True, help="Reverse output sequence of tokens.")
@click.option("--stream", is_flag=True, help="Process each line before reading the next one.")
@click.option("--dropout_prob", type=click.FLOAT, default=0, show_default=True,
              help="BPE-dropout probability (the probability of a merge being dropped)")
def encode(model, output_type, n_threads, bos, eos, reverse, stream, dropout_prob):
    """Encode text to ids or subwords."""
    if n_threads < -1 or n_threads == 0:  # yttm_cli.py:110-114
        raise ValueError('Invalid value for "--n_threads": must be -1 or positive integer, not "%d"' % n_threads)
    core = _Core(model, n_threads)
    core.encode_cli(output_type, stream, bos, eos, reverse, dropout_prob)  # the C++ loop: yttm_encode_cli (host_cli.cpp)


def _parse_ignore_ids(ctx, param, value):
    if value is None:
        return None
    try:
        return [int(v) for v in value.split(",")]
    except ValueError:
        raise click.BadParameter("Bad format: expected list of comma-separated integers, but got {}".format(value))


@main.command()
@click.option("--model", type=click.Path(exists=True), required=True, help="Path to file with learned model.")
@click.option("--ignore_ids", type=click.STRING, callback=_parse_ignore_ids, required=False,
              help="List of indices to ignore for decoding. Example: --ignore_ids=1,2,3")
def decode(model, ignore_ids):
    """Decode ids to text."""
    core = _Core(model)
    core.decode_cli(ignore_ids)


@main.command()
@click.option("--model", type=click.Path(exists=True), required=True, help="Path to file with learned model.")
@click.option("--verbose", is_flag=True, help="Add merging rules.")
def vocab(model, verbose):
    """Print list of learned subwords."""
    core = _Core(model)
    core.vocab_cli(verbose)


if __name__ == "__main__":
    main()
