#!/usr/bin/env node
'use strict';
/*
 * Headless stand-in for the reference's browser shell (src/main.ts): renders a
 * scene with HipRenderer and prints one JSON line; optionally writes the
 * presented image as a binary PPM and raw read-backs for parity checks.
 *
 *   node cli.js --scene scene.json --width 640 --height 360 --spp 16 --frames 4
 *               [--bounces 6] [--ppm out.ppm] [--dump prefix] [--per-frame]
 */
const fs = require('fs');
const { HipRenderer } = require('./hip-renderer');

function parseArgs(argv) {
    const a = { width: 1536, height: 864, spp: 1, frames: 1, bounces: 6 }; // canvas 1536x864: index.html:33
    for (let i = 2; i < argv.length; i++) {
        const k = argv[i];
        if (k === '--per-frame') { a.perFrame = true; continue; }
        const v = argv[++i];
        if (k === '--scene') a.scene = v;
        else if (k === '--width') a.width = parseInt(v, 10);
        else if (k === '--height') a.height = parseInt(v, 10);
        else if (k === '--spp') a.spp = parseInt(v, 10);
        else if (k === '--frames') a.frames = parseInt(v, 10);
        else if (k === '--bounces') a.bounces = parseInt(v, 10);
        else if (k === '--ppm') a.ppm = v;
        else if (k === '--dump') a.dump = v;
        else throw new Error('unknown argument ' + k);
    }
    if (!a.scene) throw new Error('--scene is required');
    return a;
}

async function main() {
    const a = parseArgs(process.argv);
    const canvas = { width: a.width, height: a.height, pixels: a.ppm ? new Uint8Array(a.width * a.height * 4) : null };
    const renderer = await HipRenderer.Create(canvas, a.scene, { maxBounces: a.bounces });
    if (!renderer) process.exit(1);
    renderer.setSamplesPerPixel(a.spp);
    const t0 = process.hrtime.bigint();
    if (a.perFrame) for (let f = 0; f < a.frames; f++) renderer.render();
    else renderer.renderFrames(a.frames);
    renderer.sync();
    const seconds = Number(process.hrtime.bigint() - t0) / 1e9;
    const counters = renderer.getCounters();
    const out = {
        scene: a.scene, width: a.width, height: a.height, frames: a.frames, spp_per_frame: renderer.samplesPerPixel,
        frame_index: renderer.frameIndex, rays: counters[0], seconds: seconds,
        mrays_per_s: counters[0] / seconds / 1e6, ms_per_frame: (seconds * 1e3) / a.frames,
        triangle_tests_u32: await renderer.getTriangleTestsSinceLastCheck(),
    };
    if (a.ppm) {
        const px = canvas.pixels, rgb = Buffer.alloc(a.width * a.height * 3);
        for (let i = 0, n = a.width * a.height; i < n; i++) {
            rgb[3 * i] = px[4 * i]; rgb[3 * i + 1] = px[4 * i + 1]; rgb[3 * i + 2] = px[4 * i + 2];
        }
        fs.writeFileSync(a.ppm, Buffer.concat([Buffer.from('P6\n' + a.width + ' ' + a.height + '\n255\n'), rgb]));
    }
    if (a.dump) {
        fs.writeFileSync(a.dump + '.accum.f32', Buffer.from(renderer.readAccum().buffer));
        fs.writeFileSync(a.dump + '.ids.i32', Buffer.from(renderer.readIds().buffer));
        fs.writeFileSync(a.dump + '.frame.f16', Buffer.from(renderer.readFrameF16().buffer));
    }
    renderer.destroy();
    console.log(JSON.stringify(out));
}

main().catch((e) => { console.error(e); process.exit(1); });
