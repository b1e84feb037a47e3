"""Image preparation of a training item as tensor code (reference data/zju_mocap.py:209-262 does it with OpenCV, which is
not part of this image): body-mask rim (cv2.erode / cv2.dilate), lens undistortion (cv2.undistort), resizing (cv2.resize).

PARITY UNPINNED against OpenCV -- restated from its documented behaviour:
  * erode / dilate with a k x k box kernel and the default border (pixels outside the image do not take part);
  * undistort: for every pixel of the result the distorted source position by the (k1, k2, p1, p2, k3) model, sampled
    bilinearly (OpenCV interpolates with weights quantised to 1/32; here the weights are exact: results agree to about
    half a grey level), outside the source -> 0;
  * resize: INTER_LINEAR with pixel centres at half integers and edge replication, no anti-aliasing; INTER_NEAREST takes
    source index floor(dst * src / dst_size)."""
import torch
import torch.nn.functional as F


def erode(mask, k=5):
    m = mask.float()[None, None]
    return (-F.max_pool2d(-m, k, stride=1, padding=k // 2)).reshape(mask.shape).to(mask.dtype)


def dilate(mask, k=5):
    m = mask.float()[None, None]
    return F.max_pool2d(m, k, stride=1, padding=k // 2).reshape(mask.shape).to(mask.dtype)


def rim_mask(mask_in, erode_mask=True, border=5):
    """ZJUMOCAPDataset.get_mask (zju_mocap.py:209-219): 1 on the body, 0 on the background and, when eroding, 100 on the rim
    where a border x border neighbourhood sees both."""
    mask = (mask_in != 0).to(torch.int64)
    if erode_mask:
        mask = torch.where((dilate(mask, border) - erode(mask, border)) == 1, torch.full_like(mask, 100), mask)
    return mask


def undistort(img, K, D):
    """cv2.undistort(img, K, D, None): img (H,W) or (H,W,C) float, K (3,3), D (k1, k2, p1, p2[, k3]).  New camera matrix =
    K."""
    squeeze = img.dim() == 2
    x = img[..., None] if squeeze else img
    H, W, C = x.shape
    dev = x.device
    d = list(torch.as_tensor(D, dtype=torch.float64).reshape(-1).tolist()) + [0.0] * 5
    k1, k2, p1, p2, k3 = d[:5]
    if not any((k1, k2, p1, p2, k3)):
        return img.clone()
    fx, fy, cx, cy = (float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]))
    v, u = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float64), torch.arange(W, device=dev, dtype=torch.float64),
                          indexing="ij")
    xn, yn = (u - cx) / fx, (v - cy) / fy
    r2 = xn * xn + yn * yn
    radial = 1.0 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2
    xd = xn * radial + 2.0 * p1 * xn * yn + p2 * (r2 + 2.0 * xn * xn)
    yd = yn * radial + p1 * (r2 + 2.0 * yn * yn) + 2.0 * p2 * xn * yn
    us, vs = xd * fx + cx, yd * fy + cy
    u0, v0 = torch.floor(us), torch.floor(vs)
    au, av = (us - u0).float(), (vs - v0).float()
    out = torch.zeros(H, W, C, device=dev, dtype=torch.float32)
    src = x.float()
    for dv, wv in ((0, 1.0 - av), (1, av)):
        for du, wu in ((0, 1.0 - au), (1, au)):
            uu, vv = (u0 + du).long(), (v0 + dv).long()
            ok = (uu >= 0) & (uu < W) & (vv >= 0) & (vv < H)
            val = src[vv.clamp(0, H - 1), uu.clamp(0, W - 1)]
            out += val * (wu * wv * ok.float())[..., None]
    out = out.to(img.dtype) if img.dtype.is_floating_point else out.round().to(img.dtype)
    return out[..., 0] if squeeze else out


def resize_linear(img, size):
    """cv2.resize(img, (size[1], size[0]), interpolation=cv2.INTER_LINEAR) for (H,W,C) float images."""
    x = img.permute(2, 0, 1)[None].float()
    return F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=False, antialias=False)[0].permute(1, 2, 0)


def resize_nearest(mask, size):
    """cv2.resize(mask, (size[1], size[0]), interpolation=cv2.INTER_NEAREST) for (H,W) integer masks."""
    H, W = mask.shape
    ys = torch.div(torch.arange(size[0], device=mask.device) * H, size[0], rounding_mode="floor").clamp(max=H - 1)
    xs = torch.div(torch.arange(size[1], device=mask.device) * W, size[1], rounding_mode="floor").clamp(max=W - 1)
    return mask[ys][:, xs]


def resize_area(img, size):
    """cv2.resize(img, (size[1], size[0]), interpolation=cv2.INTER_AREA) for (H,W,C) float images: box average of the source
    pixels a result pixel covers (exact for integer reduction factors; for fractional ones OpenCV weights the partially
    covered border pixels, here every result pixel averages the source pixels whose index range it spans)."""
    x = img.permute(2, 0, 1)[None].float()
    return F.interpolate(x, size=tuple(size), mode="area")[0].permute(1, 2, 0)
