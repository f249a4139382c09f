"""`get_network_prediction` -- same contract as the reference helper
(/root/reference/graph_networks/network_utils.py:4-20): call the network with keyword arguments,
print the traceback and re-raise on any failure (e.g. device out of memory)."""
import traceback


def get_network_prediction(network, x, adj_e_index, adj_e_features, col_e_idx, col_e_features=None):
    try:
        probs, *_ = network(x=x, adj_e_index=adj_e_index, adj_e_features=adj_e_features, col_e_idx=col_e_idx,
                            col_e_features=col_e_features)
    except Exception:
        print(traceback.format_exc())
        raise
    return probs
