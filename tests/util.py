import torch


def random_rois(R, n_img, H_img, W_img, seed=0, edge_cases=True):
    """(R,5) fp32 [batch, x0, y0, x1, y1] in image pixels, with the reference-relevant edge cases."""
    g = torch.Generator().manual_seed(seed)
    x0 = torch.rand(R, generator=g) * (W_img - 17)
    y0 = torch.rand(R, generator=g) * (H_img - 17)
    w = 16 + torch.rand(R, generator=g) * (min(400, W_img) - 16)
    h = 16 + torch.rand(R, generator=g) * (min(300, H_img) - 16)
    b = torch.randint(0, n_img, (R,), generator=g).float()
    rois = torch.stack([b, x0, y0, (x0 + w).clamp(max=W_img), (y0 + h).clamp(max=H_img)], 1)
    if edge_cases and R >= 8:
        rois[0, 1:] = torch.tensor([-50.0, -50.0, -10.0, -10.0])  # fully outside (top-left)
        rois[1, 1:] = torch.tensor([W_img - 10.0, H_img - 10.0, W_img + 100.0, H_img + 100.0])  # spills out
        rois[2, 1:] = torch.tensor([100.0, 100.0, 100.0, 100.0])  # zero-size -> 1x1
        rois[3, 1:] = torch.tensor([300.0, 200.0, 100.0, 50.0])  # malformed (end < start)
        rois[4, 1:] = torch.tensor([0.0, 0.0, float(W_img), float(H_img)])  # whole image
        rois[5, 1:] = torch.tensor([4.0, 4.0, 11.9, 11.9])  # sub-cell box: empty bins
        rois[6, 1:] = torch.tensor([3.5, 3.5, 12.5, 20.5])  # .5 rounding (half away from zero)
        rois[7, 1:] = torch.tensor([W_img + 50.0, H_img + 50.0, W_img + 90.0, H_img + 90.0])  # fully outside
    return rois
