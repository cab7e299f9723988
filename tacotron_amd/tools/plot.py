"""Alignment plot (reference tools/plot.py:6-16): encoder timestep against decoder timestep, saved as PNG.
matplotlib is imported on first use with the non-interactive Agg backend, as the reference selects it."""


def plot_alignment(alignment, path, info=None):
    """alignment [T_in, steps] (one utterance of `Tacotron.alignments`, device tensor or array)."""
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    if hasattr(alignment, 'cpu'):
        alignment = alignment.cpu().numpy()
    fig, ax = plt.subplots()
    image = ax.imshow(alignment, aspect='auto', origin='lower', interpolation='none')
    fig.colorbar(image, ax=ax)
    ax.set_xlabel('Decoder timestep' if info is None else 'Decoder timestep\n\n' + info)
    ax.set_ylabel('Encoder timestep')
    fig.tight_layout()
    fig.savefig(path, format='png')
    plt.close(fig)
