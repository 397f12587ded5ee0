'use strict';
/*
 * Image files of the presented frame: what the reference's accumulator pass puts on the canvas
 * (`return vec4(accumulatedColor, 1.0)`, public/accumulator.wgsl:67: 8-bit unorm, no gamma),
 * written as binary PPM or as PNG (8-bit RGB, zlib from Node, no dependencies).
 */
const fs = require('fs');
const zlib = require('zlib');

/* RGBA8 (width*height*4) -> tightly packed RGB rows */
function rgbRows(rgba, width, height) {
    const rgb = Buffer.alloc(width * height * 3);
    for (let i = 0, n = width * height; i < n; i++) {
        rgb[3 * i] = rgba[4 * i];
        rgb[3 * i + 1] = rgba[4 * i + 1];
        rgb[3 * i + 2] = rgba[4 * i + 2];
    }
    return rgb;
}

function encodePPM(rgba, width, height) {
    return Buffer.concat([Buffer.from('P6\n' + width + ' ' + height + '\n255\n'), rgbRows(rgba, width, height)]);
}

const CRC_TABLE = (() => {
    const table = new Uint32Array(256);
    for (let n = 0; n < 256; n++) {
        let c = n;
        for (let k = 0; k < 8; k++) c = (c & 1) ? (0xedb88320 ^ (c >>> 1)) : (c >>> 1);
        table[n] = c >>> 0;
    }
    return table;
})();

function crc32(buffers) {
    let c = 0xffffffff;
    for (const b of buffers) for (let i = 0; i < b.length; i++) c = CRC_TABLE[(c ^ b[i]) & 0xff] ^ (c >>> 8);
    return (c ^ 0xffffffff) >>> 0;
}

function pngChunk(type, data) {
    const head = Buffer.alloc(8), tail = Buffer.alloc(4), tag = Buffer.from(type, 'ascii');
    head.writeUInt32BE(data.length, 0);
    tag.copy(head, 4);
    tail.writeUInt32BE(crc32([tag, data]), 0);
    return Buffer.concat([head, data, tail]);
}

function encodePNG(rgba, width, height) {
    const rgb = rgbRows(rgba, width, height), stride = width * 3;
    const raw = Buffer.alloc((stride + 1) * height); // every scanline starts with filter type 0
    for (let y = 0; y < height; y++) rgb.copy(raw, y * (stride + 1) + 1, y * stride, (y + 1) * stride);
    const header = Buffer.alloc(13);
    header.writeUInt32BE(width, 0);
    header.writeUInt32BE(height, 4);
    header[8] = 8;  // bits per channel
    header[9] = 2;  // colour type: RGB
    return Buffer.concat([
        Buffer.from([0x89, 0x50, 0x4e, 0x47, 0x0d, 0x0a, 0x1a, 0x0a]),
        pngChunk('IHDR', header), pngChunk('IDAT', zlib.deflateSync(raw)), pngChunk('IEND', Buffer.alloc(0)),
    ]);
}

/* Writes `file` as PNG if its name ends in .png, as PPM otherwise. */
function writeImage(file, rgba, width, height) {
    const png = /\.png$/i.test(file);
    fs.writeFileSync(file, png ? encodePNG(rgba, width, height) : encodePPM(rgba, width, height));
}

module.exports = { encodePPM, encodePNG, writeImage };
