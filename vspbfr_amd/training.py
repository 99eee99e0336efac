"""Training-mode (differentiable) forward of `Restoration_net` over the gfx950 operators -- the generator half of the training step
(reference restoration_train.py:153-255 runs `generator(low_imgs, feats, latents, noise)` under autograd).

The inference modules (restorenet.py / layers.py) run fused, `no_grad` launches: style scaling inside the conv staging, demodulation,
noise, bias and activation in the conv epilogue -- none of which records a graph.  This module evaluates the SAME parameters
(same module tree, same state-dict keys) the way the reference's own forward does, op by op, with every op differentiable:

    convolutions      op.conv2d_gradfix.conv2d / conv_transpose2d   (data gradient on the forward kernels, weight gradient
                                                                     vsp_conv2d_wgrad_f32; the per-sample weights of a modulated
                                                                     layer go through the `groups = batch` form, as in
                                                                     models/RestoreNet.py:373-416)
    bias + leaky-ReLU op.fused_leaky_relu                            (vsp_fused_bias_act_f32, any-order autograd)
    blur / upsample   op.upfirdn2d                                   (vsp_upfirdn2d_f32, adjoint = upfirdn2d with flipped taps)
    everything else   torch tensor algebra on the device (weight modulation / demodulation, F.linear, concatenations) -- glue
                      whose gradients autograd derives; F.linear is a plain library GEMM.

`tests/test_hip_models.py::test_restorenet64_training_gradients` compares output and parameter / input gradients with the
reference's own backward pass (tests/golden/restorenet64_grad.npz, tools/make_golden.py::gen_restorenet64_grad)."""
import torch
import torch.nn.functional as F

from .op import conv2d_gradfix, fused_leaky_relu, upfirdn2d


def equal_linear(x, lin):
    """EqualLinear.forward (reference models/RestoreNet.py:161-171)."""
    if lin.activation:
        return fused_leaky_relu(F.linear(x, lin.weight * lin.scale), lin.bias * lin.lr_mul)
    return F.linear(x, lin.weight * lin.scale, bias=None if lin.bias is None else lin.bias * lin.lr_mul)


def _blur(x, blur):
    return upfirdn2d(x, blur.kernel, pad=blur.pad)


def modulated_conv(x, conv, style, modulation=None):
    """ModulatedConv2d.forward, fused branch (reference models/RestoreNet.py:373-416): per-sample weights
    w[b] = scale * W * s[b], demodulated, applied as ONE grouped convolution with groups = batch."""
    B, cin, H, W = x.shape
    k, cout = conv.kernel_size, conv.out_channel
    s = equal_linear(style, modulation if modulation is not None else conv.modulation).view(B, 1, cin, 1, 1)
    w = conv.scale * conv.weight * s                                         # (B, Cout, Cin, k, k)
    if conv.demodulate:
        w = w * torch.rsqrt(w.pow(2).sum([2, 3, 4]) + 1e-8).view(B, cout, 1, 1, 1)
    if conv.upsample:
        wt = w.transpose(1, 2).reshape(B * cin, cout, k, k)
        out = conv2d_gradfix.conv_transpose2d(x.reshape(1, B * cin, H, W), wt, padding=0, stride=2, groups=B)
        return _blur(out.view(B, cout, out.shape[2], out.shape[3]), conv.blur)
    if conv.downsample:
        x = _blur(x, conv.blur)
        H, W = x.shape[2], x.shape[3]
        out = conv2d_gradfix.conv2d(x.reshape(1, B * cin, H, W), w.view(B * cout, cin, k, k), padding=0, stride=2, groups=B)
    else:
        out = conv2d_gradfix.conv2d(x.reshape(1, B * cin, H, W), w.view(B * cout, cin, k, k), padding=conv.padding,
                                    dilation=conv.dilation, groups=B)
    return out.view(B, cout, out.shape[2], out.shape[3])


def styled_conv(x, layer, style, noise):
    """StyledConv.forward (reference models/RestoreNet.py:599-603): conv -> noise -> FusedLeakyReLU."""
    out = modulated_conv(x, layer.conv, style)
    out = out + layer.noise.weight * noise
    return fused_leaky_relu(out, layer.activate.bias)


def smart_layer(x, layer, style, noise):
    """SMART_layer.forward (reference models/RestoreNet.py:220-244): the four dilated branches share ONE modulation; fusion conv,
    FusedLeakyReLU, noise, FusedLeakyReLU."""
    outs = [modulated_conv(x, m, style, modulation=layer.modulation) for m in layer.ModulatedConv2ds]
    out = torch.cat(outs, dim=1)
    f = layer.fusion[0]
    out = conv2d_gradfix.conv2d(out, f.weight * f.scale, padding=1)
    out = fused_leaky_relu(out, layer.fusion[1].bias)
    out = out + layer.noise.weight * noise
    return fused_leaky_relu(out, layer.activate.bias)


def large_conv_layer(x, layer):
    """LargeConvLayer.forward (reference models/RestoreNet.py:773-787)."""
    outs = [conv2d_gradfix.conv2d(x, m.weight * m.scale, padding=m.padding, dilation=m.dilation) for m in layer.dilated_convs]
    f = layer.fusion[0]
    out = conv2d_gradfix.conv2d(torch.cat(outs, dim=1), f.weight * f.scale, padding=f.padding)
    out = fused_leaky_relu(out, layer.fusion[1].bias)
    return fused_leaky_relu(out, layer.activate.bias)


def to_rgb(x, layer, style, skip=None):
    """ToRGB.forward (reference models/RestoreNet.py:657-666)."""
    out = modulated_conv(x, layer.conv, style) + layer.bias
    if skip is not None:
        u = layer.upsample
        out = out + upfirdn2d(skip, u.kernel, up=u.factor, down=1, pad=u.pad)
    return out


def restoration_net_forward(net, images, de_feats, pre_styles, noise_styles, enc_noise, dec_noise, inject_index=None):
    """Restoration_net.forward + encoder_forward (reference models/RestoreNet.py:915-942, 968-1046) with explicit noise maps
    (enc_noise: 2 per encoder level, dec_noise: 1 + 2 per decoder level, as the inference module takes them).  Dropout2d of
    `final_linear` follows the module's mode (identity in eval)."""
    B = images.shape[0]
    styles = []
    for z in noise_styles:
        h = z * torch.rsqrt(torch.mean(z ** 2, dim=1, keepdim=True) + 1e-8)                 # PixelNorm
        for lin in list(net.style)[1:]:
            h = equal_linear(h, lin)
        styles.append(h)
    if len(styles) < 2:
        noise_latent = styles[0].unsqueeze(1).repeat(1, net.n_latent, 1)
    else:
        if inject_index is None:
            raise RuntimeError("two noise codes need an explicit inject_index")
        noise_latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                                  styles[1].unsqueeze(1).repeat(1, net.n_latent - inject_index, 1)], 1)
    latent = torch.cat([pre_styles[:, :net.n_latent], noise_latent], dim=-1)
    latent_cp = torch.flip(latent, dims=[1])

    out = large_conv_layer(images, net.down_from_big)
    feats = []
    for ii in range(0, len(net.encoder_convs), 2):
        out = smart_layer(out, net.encoder_convs[ii], latent_cp[:, ii], enc_noise[ii])
        feats.append(out)
        out = styled_conv(out, net.encoder_convs[ii + 1], latent_cp[:, ii], enc_noise[ii + 1])
    out = large_conv_layer(out, net.final_layer)
    x_global = net.final_linear[1](equal_linear(out.reshape(B, -1), net.final_linear[0]))
    early = equal_linear(x_global, net.final_transfer)
    feats.append(out + early.view(B, -1, 4, 4))
    feats = feats[::-1]

    def sty(i):
        return torch.cat([latent[:, i], x_global], dim=1)

    out = smart_layer(feats[0], net.conv1, sty(0), dec_noise[0])
    skip = to_rgb(out, net.to_rgb1, sty(1))
    i = 1
    for j in range(net.log_size - 2):
        out = styled_conv(out, net.convs[2 * j], sty(i), dec_noise[1 + 2 * j])
        k = (i + 1) // 2
        out = out + feats[k] + de_feats[k]
        out = smart_layer(out, net.convs[2 * j + 1], sty(i + 1), dec_noise[2 + 2 * j])
        skip = to_rgb(out, net.to_rgbs[j], sty(i + 2), skip)
        i += 2
    return skip

